// dr_brick.h -- geometry shared by the brick-centric march kernels (march_flat.hip, ray_passes.hip).
//
// The volume's cells (one cell = the unit cube between 8 voxels, indexed by the low voxel) are grouped
// into bricks of BRK^3 cells. A sample belongs to the brick that holds the cell of its centre tap; the
// six normal taps sit 1e-3 world units away (VR.py:193), which is < 1 voxel whenever max(dim)-1 < 2000,
// so they touch at most the neighbouring cell: the LDS box of a brick is BRK+3 voxels wide
// (one voxel below, two above).
//
// Layer of a brick FOR A RAY = Manhattan distance (in bricks) from the brick that holds the ray's FIRST sample.
// Along a line every coordinate is monotone, so the bricks a ray visits have strictly increasing layers:
// per-(ray, layer) partial composites can be combined front to back without knowing which brick produced
// them, and a ray needs at most NBx+NBy+NBz-2 layers. This holds wherever the ray starts -- also for a camera
// inside the volume, whose rays begin BEHIND the eye (the reference does not clip tmin at 0, VR.py:28-53).
// (Each brick also has a camera-based layer -- distance from the camera's brick -- used only to run the alpha
// pre-pass front to back in groups of bricks; for a camera outside the volume it is the ray's layer plus a per-ray
// constant.)
#pragma once
#include "dr_device.h"

namespace dr {

#ifndef DR_BRK
#define DR_BRK 12
#endif
constexpr int BRK = DR_BRK;                    // cells per brick edge
constexpr int BOX = BRK + 3;               // voxels per LDS box edge
// LDS strides of the box (z fastest). All three are odd, and SX mod 32 differs from SZ = 1 and from SY mod 32, so
// that the ~10 distinct cells a 32-lane group touches while walking along a ray land in distinct banks
// (the orbit cameras march mostly in the x-z plane: SX = 225 = 1 (mod 32) aliased x-steps with z-steps).
#ifndef DR_BOX_PADX
#define DR_BOX_PADX 12
#endif
#ifndef DR_BOX_PADY
#define DR_BOX_PADY 0
#endif
constexpr int BOX_SY = BOX + DR_BOX_PADY;
constexpr int BOX_SX = BOX * BOX_SY + DR_BOX_PADX;
constexpr int BOX_VOX = BOX * BOX * BOX;   // voxels staged per brick
constexpr int BOX_LDS = BOX * BOX_SX;      // LDS elements reserved for them
constexpr float BRICK_EPS = 5e-5f;         // world-space slack of the conservative ray/brick tests (float error: ~2e-7)

struct BrickGrid {
    int NBx, NBy, NBz, NL;  // bricks per axis, number of layers (NBx+NBy+NBz-2)
};

__host__ __device__ inline BrickGrid make_brick_grid(int VX, int VY, int VZ) {
    BrickGrid g;
    g.NBx = (VX - 1 + BRK - 1) / BRK; g.NBy = (VY - 1 + BRK - 1) / BRK; g.NBz = (VZ - 1 + BRK - 1) / BRK;
    g.NL = g.NBx + g.NBy + g.NBz - 2;
    return g;
}

// The canonical per-axis coordinate of VR.py:163-168 (must stay bit-identical everywhere it is used:
// it decides which brick owns a sample).
__device__ __forceinline__ void axis_coord(float pos, float sc, int &cell, float &fr) {
    const float q = fminf(1.0f, fmaxf(0.0f, fmaf(0.5f, pos, 0.5f))) * sc;
    // q >= 0: q - floor(q) is exact, so v_fract_f32 returns the same bits; the conversion truncates = floor
    fr = __builtin_amdgcn_fractf(q);
    cell = (int)q;
}

// brick coordinate of the camera along one axis (unclamped linear map, may be negative or >= NB)
__device__ __forceinline__ int cam_brick(float cam, float sc) {
    return (int)floorf(fmaf(0.5f, cam, 0.5f) * sc * (1.0f / BRK));
}
__device__ __forceinline__ int axis_layer_min(int cb, int NB) { return cb < 0 ? -cb : (cb > NB - 1 ? cb - (NB - 1) : 0); }

// A ray is "regular" when the brick pipeline handles it; the others -- single-sample rays, whose positions are 0/0 in the
// reference (VR.py:279-280) -- are marched whole by the per-ray pass. Must be evaluated identically in every pass.
__device__ __forceinline__ bool ray_is_regular(int n) { return n >= 2; }

// Brick that holds sample 0 of a ray (position cam + t0 * vd: mix(t0, exit, 0) == t0 exactly, VR.py:277-280), from the
// canonical cell computation.
__device__ __forceinline__ void entry_brick(float scx, float scy, float scz, f3 cam, f3 vd, float t0, int &ebx, int &eby,
                                            int &ebz) {
    float fr;
    int x0, y0, z0;
    axis_coord(fmaf(t0, vd.x, cam.x), scx, x0, fr);
    axis_coord(fmaf(t0, vd.y, cam.y), scy, y0, fr);
    axis_coord(fmaf(t0, vd.z, cam.z), scz, z0, fr);
    ebx = x0 / BRK; eby = y0 / BRK; ebz = z0 / BRK;
}
__device__ __forceinline__ int ray_layer(int bx, int by, int bz, int ebx, int eby, int ebz) {
    return abs(bx - ebx) + abs(by - eby) + abs(bz - ebz);
}
// camera-based layer of a brick (phases of the alpha pre-pass): distance from the camera's (unclamped) brick
__device__ __forceinline__ int camera_layer(int bx, int by, int bz, f3 cam, float scx, float scy, float scz, int NBx, int NBy,
                                            int NBz) {
    const int cbx = cam_brick(cam.x, scx), cby = cam_brick(cam.y, scy), cbz = cam_brick(cam.z, scz);
    const int lmin = axis_layer_min(cbx, NBx) + axis_layer_min(cby, NBy) + axis_layer_min(cbz, NBz);
    return abs(bx - cbx) + abs(by - cby) + abs(bz - cbz) - lmin;
}

// trilinear tap from the LDS box; identical arithmetic to tri_sample (x -> y -> z lerps)
__device__ __forceinline__ float tri_lds(const float *box, int base, float fx, float fy, float fz) {
    float a = mixf(box[base], box[base + BOX_SX], fx);
    float b = mixf(box[base + BOX_SY], box[base + BOX_SX + BOX_SY], fx);
    float zl = mixf(a, b, fy);
    a = mixf(box[base + 1], box[base + BOX_SX + 1], fx);
    b = mixf(box[base + BOX_SY + 1], box[base + BOX_SX + BOX_SY + 1], fx);
    float zh = mixf(a, b, fy);
    return mixf(zl, zh, fz);
}

}  // namespace dr
