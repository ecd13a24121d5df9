"""differender_amd -- MI355X-native differentiable volume raycaster (HIP/gfx950) behind the
Python API of nanovis/Differender. See DESIGN.md."""
__version__ = "0.1.0"
