// ray_setup.hip -- ray generation, box clipping, sample count, jitter (gfx950).
// Replaces VolumeRaycaster.compute_entry_exit (VR.py:221-259), get_ray_direction (VR.py:127-151) and
// get_entry_exit_points (VR.py:28-53) of the reference.
//
// This translation unit is compiled with -ffp-contract=off so that the sample count
// n = floor(sr * len * diag) + 1 is bit-identical to the CPU oracle: a one-ulp difference in the
// product would flip n for some pixel and every downstream comparison with it.  The kernel is
// negligible in time (24 B written per pixel), so strict IEEE arithmetic costs nothing that matters.
#include "dr_device.h"
#include "dr_kernels.h"

namespace dr {

struct RaySetupParams {
    const float *cam;  // [views][3]
    int n_views, W, H;
    int img_W, row0;   // band: buffer row i is image row row0 + i of img_W
    float near_, near_w, near_h, vol_diag, sr;
    uint32_t jitter_seed, view_base;
    float *entry, *exit_, *rays;
    int32_t *nsamp;
};

__device__ __forceinline__ f3 cross3(f3 a, f3 b) {
    return make_f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

// One thread per pixel; a wave covers an 8x8 pixel tile (the reference's image tile, VR.py:104-112).
__global__ __launch_bounds__(256) void ray_setup_kernel(RaySetupParams P) {
    const int tiles_j = (P.H + 7) >> 3;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const int i = (wave / tiles_j) * 8 + (lane >> 3);
    const int j = (wave % tiles_j) * 8 + (lane & 7);
    const int view = blockIdx.y;
    if (i >= P.W || j >= P.H) return;

    const float *cam = P.cam + 3 * view;
    const f3 lf = make_f3(cam[0], cam[1], cam[2]);
    const f3 view_dir = normalized3(make_f3(-lf.x, -lf.y, -lf.z));
    const float x = ((float)(i + P.row0) + 0.5f) / (float)P.img_W;
    const float y = ((float)j + 0.5f) / (float)P.H;
    const float u = x - 0.5f, v = y - 0.5f;
    f3 up = make_f3(0.f, 1.f, 0.f);
    const f3 right = normalized3(cross3(view_dir, up));
    up = normalized3(cross3(right, view_dir));
    const f3 near_m = make_f3(lf.x + P.near_ * view_dir.x, lf.y + P.near_ * view_dir.y, lf.z + P.near_ * view_dir.z);
    const float uw = u * P.near_w, vh = v * P.near_h;
    const f3 near_pos = make_f3((near_m.x + uw * right.x) + vh * up.x, (near_m.y + uw * right.y) + vh * up.y,
                                (near_m.z + uw * right.z) + vh * up.z);
    const f3 vd = normalized3(make_f3(near_pos.x - lf.x, near_pos.y - lf.y, near_pos.z - lf.z));

    // slab test against [-1,1]^3
    const float fx = 1.0f / vd.x, fy = 1.0f / vd.y, fz = 1.0f / vd.z;
    const float t1 = (-1.0f - lf.x) * fx, t2 = (1.0f - lf.x) * fx;
    const float t3 = (-1.0f - lf.y) * fy, t4 = (1.0f - lf.y) * fy;
    const float t5 = (-1.0f - lf.z) * fz, t6 = (1.0f - lf.z) * fz;
    float tmin = fmaxf(fmaxf(fminf(t1, t2), fminf(t3, t4)), fminf(t5, t6));
    const float tmax = fminf(fminf(fmaxf(t1, t2), fmaxf(t3, t4)), fmaxf(t5, t6));
    const bool hit = !(tmax < 0.0f || tmin > tmax);

    const float ray_len = tmax - tmin;
    const float n_samples = (hit ? 1.0f : 0.0f) * (floorf(P.sr * ray_len * P.vol_diag) + 1.0f);
    const size_t p = ((size_t)view * P.W + i) * P.H + j;
    if (P.jitter_seed != 0u) {
        const float uu = jitter_u(P.jitter_seed, P.view_base + (uint32_t)view, (uint32_t)((i + P.row0) * P.H + j));
        tmin += uu * ray_len / n_samples;
    }
    P.entry[p] = tmin;
    P.exit_[p] = tmax;
    P.rays[3 * p + 0] = vd.x;
    P.rays[3 * p + 1] = vd.y;
    P.rays[3 * p + 2] = vd.z;
    P.nsamp[p] = (int32_t)n_samples;
}

hipError_t launch_ray_setup(const float *cam, int n_views, int W, int H, int img_W, int row0, int VX, int VY, int VZ,
                            double fov_rad, double near_plane, float sr, uint32_t jitter_seed, uint32_t view_base,
                            float *entry, float *exit_, float *rays, int32_t *nsamp, hipStream_t stream) {
    RaySetupParams P;
    // VR.py:146-147: Python doubles, then rounded once (ti.tan of a Python float is math.tan)
    const double near_h = 2.0 * tan(fov_rad) * near_plane;
    const double near_w = near_h * ((double)img_W / (double)H);
    P.cam = cam; P.n_views = n_views; P.W = W; P.H = H; P.img_W = img_W; P.row0 = row0;
    P.near_ = (float)near_plane; P.near_w = (float)near_w; P.near_h = (float)near_h;
    P.vol_diag = (float)sqrt((double)(VX - 1) * (VX - 1) + (double)(VY - 1) * (VY - 1) + (double)(VZ - 1) * (VZ - 1));
    P.sr = sr; P.jitter_seed = jitter_seed; P.view_base = view_base;
    P.entry = entry; P.exit_ = exit_; P.rays = rays; P.nsamp = nsamp;
    const int tiles = ((W + 7) / 8) * ((H + 7) / 8);
    dim3 grid((tiles + 3) / 4, n_views);
    hipLaunchKernelGGL(ray_setup_kernel, grid, dim3(256), 0, stream, P);
    return hipGetLastError();
}

}  // namespace dr
