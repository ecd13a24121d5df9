from differender_amd.utils import get_tf, in_circles, get_rand_pos  # noqa: F401
