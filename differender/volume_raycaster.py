from differender_amd.volume_raycaster import VolumeRaycaster, RaycastFunction, Raycaster  # noqa: F401
