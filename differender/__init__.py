"""Import-compatibility alias: `from differender.volume_raycaster import Raycaster` and
`from differender.utils import get_tf, in_circles, get_rand_pos` (examples/test_opt_tf.py:10-11 of the
reference) resolve to the MI355X implementation in `differender_amd`."""
__version__ = "0.0.3"
