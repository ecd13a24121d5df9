from .utils import get_tf, in_circles, get_rand_pos, tex_from_pts
from .losses import ssim2d, dssim_mse_loss
