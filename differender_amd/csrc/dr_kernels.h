// dr_kernels.h -- host-side launch entry points of the kernel translation units (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dr_experiment.h"

namespace dr {

// Everything a march launch needs; filled by capi.cpp from the C-ABI arguments.
struct MarchArgs {
    const void *vol; int vol_dtype; int VX, VY, VZ; int64_t sx, sy, sz, vol_vs;
    const float *tf; int R; int64_t tf_vs;
    const float *cam, *entry, *exit_, *rays; const int32_t *nsamp;
    int n_views, W, H, S; float sr; int mode;
    int img_W, row0;  // the W rows of the buffers are rows [row0, row0 + W) of an image img_W rows wide (bands)
    float *out; int32_t *steps;
    // backward only
    const float *grad_out; const float *out_fwd;
    float *d_vol; int64_t dsx, dsy, dsz, dvol_vs;
    float *d_tf; int64_t dtf_vs;
    // brick path
    double fov_rad, near_plane;
    void *workspace; size_t workspace_bytes;
    const uint8_t *only_flagged;  // baseline backward: restrict to rays with a non-zero flag (may be null)
    const unsigned int *ws_mark;  // ... unless *ws_mark != ws_mark_expect (the workspace is not this call's forward's: the
    unsigned int ws_mark_expect;  //     flags are garbage, every ray is marched); may be null
    const unsigned int *ws_aux;   // ... or *ws_aux != ws_aux_expect (the TF-only backward over the tape: the forward left no tape of
    unsigned int ws_aux_expect;   //     this stride); may be null
    int hints;                    // DR_HINT_* bits of the forward call
    int use_live;                 // forward: per-ray live sample counts are available (alpha pre-pass)
    int pp_l0, pp_l1, pp_first;   // alpha pre-pass phase: brick layers [pp_l0, pp_l1); pp_first: no earlier phase
};

hipError_t launch_ray_setup(const float *cam, int n_views, int W, int H, int img_W, int row0, int VX, int VY, int VZ,
                            double fov_rad, double near_plane, float sr, uint32_t jitter_seed, uint32_t view_base,
                            float *entry, float *exit_, float *rays, int32_t *nsamp, hipStream_t stream);

// Plain one-lane-per-ray kernels (DR_VARIANT_BASELINE): direct global gathers, global float atomics.
int launch_march_fwd_baseline(const MarchArgs &a, hipStream_t stream);
int launch_march_bwd_baseline(const MarchArgs &a, hipStream_t stream);

// Brick-centric kernels (DR_VARIANT_AUTO): LDS-staged bricks, per-(ray,layer) partial composites.
bool brick_path_supported(int VX, int VY, int VZ, int R);
bool brick_image_supported(int W, int H, int VX, int VY, int VZ);  // [layer][pixel] slot indices stay below 2^31
size_t brick_workspace_bytes(int n_views, int W, int H, int VX, int VY, int VZ);
int tape_stride_for(int VX, int VY, int VZ, float sr, int max_samples);   // samples per ray the DR_TAPE_TF tape reserves
size_t brick_workspace_bytes_tape(int n_views, int W, int H, int VX, int VY, int VZ, int max_samples, float sr);
int launch_tf_tape_bwd(const MarchArgs &a, hipStream_t stream);     // TF-only backward over the per-sample tape (tf_tape.hip)
int launch_ray_compose(const MarchArgs &a, hipStream_t stream);      // F2
int launch_ray_exact(const MarchArgs &a, hipStream_t stream);        // F3: the rays F2 listed, sample by sample
int launch_ray_exact_bwd(const MarchArgs &a, hipStream_t stream);   // B3: the backward of the rays F3 recomputed
int launch_dtf_commit(const MarchArgs &a, hipStream_t stream);      // the backward's double d_tf table -> the caller's float tensor
int launch_ray_alpha(const MarchArgs &a, hipStream_t stream);        // alpha pre-pass: per-ray composition of one phase
int launch_ray_cross(const MarchArgs &a, hipStream_t stream);        // alpha pre-pass: exact termination sample of crossing rays
bool flat_strides_ok(int64_t sx, int64_t sy, int64_t sz);  // 32-bit in-box offsets
// Word of the workspace header that says whose coarse tape the workspace holds (a fingerprint of the forward call that
// filled it, see ws_fingerprint); a forward served by the baseline kernels clears it (flat_invalidate_workspace).
constexpr int WS_MARK_WORD = 3;
hipError_t flat_invalidate_workspace(void *workspace, size_t workspace_bytes, hipStream_t stream);
int launch_march_fwd_flat(const MarchArgs &a, hipStream_t stream);   // one lane per sample
int launch_march_bwd_flat(const MarchArgs &a, hipStream_t stream);

// Loss / optimiser epilogue (epilogue.hip)
hipError_t launch_mse_loss_grad(const float *out, const float *ref, int64_t n, float inv_norm, float *grad,
                                double *loss, hipStream_t stream);
hipError_t launch_tf_momentum_step(float *tf, const float *g, float *mom, int n, float lr, float gamma,
                                   float max_grad, hipStream_t stream);

}  // namespace dr
