/*
 * differender_hip.h -- C ABI of the MI355X (gfx950) volume-raycaster hot path.
 *
 * The reference (nanovis/Differender) has no FFI: its device code is Taichi-JIT'd Python inside
 * differender/volume_raycaster.py ("VR.py").  Each entry point below replaces one Taichi kernel (or
 * kernel pair) that VR.py's RaycastFunction / Raycaster launch; the reference lines are cited per
 * function.  INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *  - The library is stateless and re-entrant.  It allocates nothing; every buffer is owned by the
 *    caller (PyTorch on the Python side).  All pointers are DEVICE pointers of ONE device; each call runs on
 *    the device that owns them (hipPointerGetAttributes), whatever the calling thread's current device is --
 *    autograd runs backward on another thread than forward -- and restores the current device afterwards.
 *  - Work is enqueued on `stream` (a hipStream_t passed as void*); no call synchronises.
 *  - Return value: 0 on success, otherwise a hipError_t value (>0) or a DR_E* code (<0);
 *    dr_error_string() renders both.
 *  - Volume = the reference's field index space (i,j,k) of extent (VX,VY,VZ) (VR.py:481 calls it
 *    (W,D,H) of the user's (1,D,H,W) tensor), addressed with ELEMENT strides (sx,sy,sz) so the
 *    permuted view of VR.py:566,571 needs no copy.  vol_dtype: DR_F32 or DR_F16 (fp16 storage,
 *    f32 arithmetic; an extension, the reference is f32-only).
 *  - Image buffers are contiguous [view][W][H](C), exactly the (W,H,4) tensors VR.py:417,438 return.
 *  - n_views > 1 runs that many independent views in one launch (replaces the Python loops of
 *    VR.py:418-426,450-464).  *_view_stride are ELEMENT strides between views; 0 shares the buffer
 *    between all views (a shared volume then receives ONE accumulated d_vol).
 */
#ifndef DIFFERENDER_HIP_H
#define DIFFERENDER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DR_ABI_VERSION 9

enum { DR_F32 = 0, DR_F16 = 1 };
enum { DR_MODE_DIFF = 0, DR_MODE_NONDIFF = 1 };
/* kernel variant selector: AUTO picks the fastest validated kernels for the problem (brick-centric, one lane
 * per sample); BASELINE forces the plain one-lane-per-ray kernels (kept for differential testing and as the
 * fallback for problems the fast kernels do not serve). */
enum { DR_VARIANT_AUTO = 0, DR_VARIANT_BASELINE = 1 };
/* Caller hints for dr_march_fwd[_rows], OR-ed into `variant` (bits 8 and up; the variant proper is the low byte). A hint
 * only chooses between ways of computing the SAME result -- a wrong one costs time, never correctness:
 *   DR_HINT_NO_EARLY_TERMINATION  "no ray can reach alpha 0.99 with this TF": the launches of the alpha pre-pass (which the
 *       device would gate off anyway, once it has looked at the TF's largest alpha) are not issued at all. The device still
 *       checks; if the hint was wrong, every ray of the view is marched whole by the per-ray kernels (exact early termination,
 *       10-40 x slower) and workspace header word 8 counts the views it happened to.
 *   DR_HINT_EARLY_TERMINATION     "many rays terminate early": the alpha pre-pass runs front to back in groups of brick
 *       layers also below sampling rate 3, so that later groups skip the rays that are already opaque.
 * differender_amd.functional derives both from the TF tensor (largest alpha, cached per tensor version, no host sync).
 *   DR_COUNT_EVALUATED (dr_march_fwd[_rows] AND dr_march_bwd[_rows]; measurement only, bench.py): the brick kernels add up the
 *       samples whose taps they actually EVALUATED -- the work-skipping paths (empty bricks, unlit segments, dead samples behind a
 *       termination point) evaluate nothing for samples that still count as marched -- in 64-bit words of the workspace header:
 *       words 58/59 alpha pre-pass, 60/61 colour march (zeroed by the forward), 62/63 backward (accumulates until the next
 *       forward). One atomic per wave: measurably slower on scenes with many short workgroups, so never set in a timed step. */
/*   DR_TAPE_TF (dr_march_fwd[_rows] with mode DR_MODE_DIFF, and the dr_march_bwd[_rows] of the same inputs with d_vol == NULL):
 *       the caller wants the gradient w.r.t. the transfer function ONLY (BASELINE config C3; the reference's TF optimisation,
 *       examples/taichi_volume_raycaster.py). The forward then leaves a per-SAMPLE tape of (intensity, lighting term) -- 8 B
 *       per marched sample, the two things of a sample the TF gradient needs from the volume -- behind the ordinary workspace
 *       (dr_workspace_bytes_tape), and the backward is a per-ray pass over that tape: no brick is staged, no tap is taken
 *       again. Same results as without the flag (a choice between ways of computing the same thing); the backward checks on
 *       the device that the workspace holds this forward's tape and marches the rays one by one if it does not. A ray with more
 *       samples than the tape reserves per ray (ray buffers made for a higher sampling rate than this call's) never leaves its
 *       slot: forward and backward march it with the per-ray kernels (counted in header word 2). */
enum { DR_HINT_NO_EARLY_TERMINATION = 0x100, DR_HINT_EARLY_TERMINATION = 0x200, DR_COUNT_EVALUATED = 0x400, DR_TAPE_TF = 0x800 };

enum {
    DR_EINVAL = -1,      /* bad argument (null pointer, non-positive extent, unknown enum) */
    DR_EUNSUPPORTED = -2, /* valid request this build cannot serve (e.g. RCCL not present, LDS opt-in refused by the driver) */
    DR_ECOLLECTIVE = -3   /* RCCL reported an error */
};

/* DR_ABI_VERSION for the shipped kernels. A library in which any translation unit was compiled with a timing-only what-if
 * switch (kernels that compute WRONG results on purpose: tools/README.md, csrc/dr_experiment.h) answers -DR_ABI_VERSION, so a
 * loader that checks the version cannot take it for the product by accident. */
int dr_abi_version(void);
/* Bit mask of how this library was built: 0 = the shipped kernels; 1 = a what-if build with wrong results;
 * 2 = diagnostic instrumentation (per-phase clocks / counters in the workspace header; results unchanged, slower);
 * 4 = built WITHOUT one of the two tuned -mllvm compiler flags because this compiler does not know it (csrc/Makefile's probe:
 *     results unchanged, kernels a few per cent slower -- a bench line from such a build says so). */
int dr_build_flags(void);
const char *dr_error_string(int code);

/* Ray generation + box clipping + sample count + jitter.
 * Replaces VolumeRaycaster.compute_entry_exit (VR.py:221-259) incl. get_ray_direction (VR.py:127-151)
 * and get_entry_exit_points (VR.py:28-53).
 *   cam     [n_views][3] camera positions (look_from); the camera looks at the origin (VR.py:233)
 *   fov_rad, near_plane: doubles, as VolumeRaycaster.__init__ holds them (VR.py:77-78)
 *   jitter_seed: 0 = no jitter; otherwise tmin += U[0,1)*len/n with U from a counter-based hash of
 *                (seed, view_base+view, pixel) -- replaces ti.random (VR.py:255) so that forward and
 *                backward see the same offsets
 *   entry, exit_ [n_views][W][H] f32; rays [n_views][W][H][3] f32; nsamp [n_views][W][H] i32 */
int dr_ray_setup(const float *cam, int n_views, int W, int H, int VX, int VY, int VZ,
                 double fov_rad, double near_plane, float sampling_rate,
                 uint32_t jitter_seed, uint32_t view_base,
                 float *entry, float *exit_, float *rays, int32_t *nsamp, void *stream);

/* Scratch memory the fast (brick-centric) march kernels need for n_views views, in bytes; 0 when this
 * problem is only served by the baseline kernels (volume edge > 2000 voxels, or a TF of more than 2030 entries: the
 * brick kernels keep the TF, 16 B per entry, and its double-precision gradient table, 32 B per entry, in LDS beside the
 * brick). The baseline kernels have NO limit on the TF resolution (like the reference): they stage the TF and the
 * double-precision d_tf table in LDS up to 3392 entries, the TF alone up to 10176 (d_tf then accumulates with float
 * atomics on the caller's tensor), and read a larger TF where it lies.
 * The caller allocates it (device memory, 256-byte aligned), passes it to dr_march_fwd and, unchanged,
 * to the dr_march_bwd of the same inputs: the forward leaves the per-segment composite prefixes and the
 * per-ray live sample counts there (the "coarse tape", ~22 B per ray per brick layer; the backward also sums d_tf there, in
 * double, before it hands the totals to the caller's tensor). Replaces the
 * reference's render_tape field (VR.py:82,102-103: 16 B per ray per SAMPLE, twice with its gradient). */
size_t dr_workspace_bytes(int n_views, int W, int H, int VX, int VY, int VZ, int R);
/* ... the same plus the per-sample tape of a DR_TAPE_TF forward: 8 B x min(max_samples, longest possible ray at this sampling
 * rate) per ray (fixed stride: 6.4 GB for a 512^2 view of a 512^3 volume at rate 1 -- the reference's render_tape is 16 B per
 * sample, twice with its gradient: VR.py:82,102-103,116). 0 where dr_workspace_bytes() is 0. */
size_t dr_workspace_bytes_tape(int n_views, int W, int H, int VX, int VY, int VZ, int R, int max_samples, float sampling_rate);

/* Forward march: trilinear sampling, 1-D TF lookup, Phong shading, front-to-back compositing with
 * early termination at A >= 0.99.
 *   mode DR_MODE_DIFF    replaces clear_framebuffer + raycast + get_final_image
 *                        (VR.py:374-382, 261-306, 363-372); marches min(n, max_samples) samples
 *   mode DR_MODE_NONDIFF replaces raycast_nondiff + get_final_image_nondiff (VR.py:308-361)
 *   tf      [n_views or 1][R][4] f32
 *   out_rgba [n_views][W][H][4] f32 (overwritten)
 *   steps   [n_views][W][H] i32, nullable: samples that passed the termination test
 *           (= valid_sample_step_count - 1, VR.py:303,381)
 *   fov_rad, near_plane: the pinhole model the ray buffers were generated with (dr_ray_setup); the fast
 *           path uses it to find the pixels a brick projects to. Rays that do not follow the model are
 *           detected (sample-count check) and marched individually, so results stay correct.
 *   workspace: dr_workspace_bytes() bytes, or NULL to force the baseline kernels. */
int dr_march_fwd(const void *vol, int vol_dtype, int VX, int VY, int VZ,
                 int64_t sx, int64_t sy, int64_t sz, int64_t vol_view_stride,
                 const float *tf, int R, int64_t tf_view_stride,
                 const float *cam, const float *entry, const float *exit_, const float *rays,
                 const int32_t *nsamp, int n_views, int W, int H, int max_samples,
                 float sampling_rate, double fov_rad, double near_plane, int mode, int variant,
                 float *out_rgba, int32_t *steps, void *workspace, size_t workspace_bytes, void *stream);

/* Backward of the DR_MODE_DIFF march w.r.t. the volume and the transfer function: the hand-derived,
 * tape-free equivalent of get_final_image.grad + raycast.grad (Taichi autodiff, VR.py:460-461,470-471).
 *   grad_out [n_views][W][H][4]  upstream gradient of out_rgba
 *   out_rgba [n_views][W][H][4]  the forward result for the same inputs (saved by the caller)
 *   d_vol    f32, element strides (dsx,dsy,dsz), nullable; ACCUMULATED into (caller zeroes)
 *   d_tf     [n_views or 1][R][4] f32, nullable; ACCUMULATED into (caller zeroes)
 *   workspace: the buffer the forward call of the same inputs filled (fast path), or NULL (baseline).
 * Gradients w.r.t. camera and sampling rate are not defined (the reference returns None, VR.py:465). */
int dr_march_bwd(const void *vol, int vol_dtype, int VX, int VY, int VZ,
                 int64_t sx, int64_t sy, int64_t sz, int64_t vol_view_stride,
                 const float *tf, int R, int64_t tf_view_stride,
                 const float *cam, const float *entry, const float *exit_, const float *rays,
                 const int32_t *nsamp, int n_views, int W, int H, int max_samples,
                 float sampling_rate, double fov_rad, double near_plane, int variant,
                 const float *grad_out, const float *out_rgba,
                 float *d_vol, int64_t dsx, int64_t dsy, int64_t dsz, int64_t dvol_view_stride,
                 float *d_tf, int64_t dtf_view_stride,
                 void *workspace, size_t workspace_bytes, void *stream);

/* Which kernels dr_march_bwd[_rows] would run for these arguments, without running anything (no GPU needed):
 * DR_VARIANT_AUTO = the brick-centric kernels, DR_VARIANT_BASELINE = the plain ones (requested, no workspace, a
 * TF too large for LDS, a volume edge > 2000, strides beyond 32-bit in-box offsets, or more [layer][pixel] slots than
 * 32-bit indices hold: n_views, W, H as in the march call). The ONE function that decides: dr_march_bwd_rows asks it.
 * The brick-centric backward sanitises its gradients itself: a NaN adjoint (NaN pixel of grad_out, NaN voxel)
 * contributes nothing, +-inf and magnitudes beyond 1e30 are clamped per sample, a brick's flush to +-3e38 -- so its
 * caller may skip the torch.nan_to_num of VR.py:463-475. The float atomics that combine bricks, views and work items
 * saturate at +-FLT_MAX (round 4: an addend beyond 1e30 takes a clamping compare-and-swap, atomic_add_sat): where the
 * reference's nan_to_num would turn an overflow into +-3.4e38, so does this path, and no element is ever +-inf or NaN --
 * also when the call is served by the per-ray second pass alone (a stale workspace, a repaired wrong hint): that pass
 * sanitises its adjoints whenever it runs on behalf of the brick-centric backward. The plain kernels (DR_VARIANT_BASELINE)
 * propagate NaN exactly like the reference and rely on nan_to_num. (dsx,dsy,dsz) are ignored if !has_dvol. */
int dr_march_bwd_variant(int n_views, int W, int H, int VX, int VY, int VZ, int R, int64_t sx, int64_t sy, int64_t sz,
                         int64_t dsx, int64_t dsy, int64_t dsz, int has_dvol, int variant, int has_workspace);

/* Image bands: the same three calls for rows [row0, row0 + W) of an image that is img_W rows wide (SURVEY 8(e):
 * "single view => split the image into G tile bands", one band per GPU). All [n_views][W][H] buffers hold the band
 * only; pixel (i, j) of the band is pixel (row0 + i, j) of the image -- same ray, same jitter value, bit for bit, as in
 * a whole-image call. Gradients of the bands add up to the whole image's (all-reduce them like those of views).
 * The plain entry points above are these with img_W = W, row0 = 0. */
int dr_ray_setup_rows(const float *cam, int n_views, int W, int H, int img_W, int row0, int VX, int VY, int VZ,
                      double fov_rad, double near_plane, float sampling_rate,
                      uint32_t jitter_seed, uint32_t view_base,
                      float *entry, float *exit_, float *rays, int32_t *nsamp, void *stream);
int dr_march_fwd_rows(const void *vol, int vol_dtype, int VX, int VY, int VZ,
                      int64_t sx, int64_t sy, int64_t sz, int64_t vol_view_stride,
                      const float *tf, int R, int64_t tf_view_stride,
                      const float *cam, const float *entry, const float *exit_, const float *rays,
                      const int32_t *nsamp, int n_views, int W, int H, int max_samples,
                      float sampling_rate, double fov_rad, double near_plane, int mode, int variant,
                      float *out_rgba, int32_t *steps, void *workspace, size_t workspace_bytes,
                      int img_W, int row0, void *stream);
int dr_march_bwd_rows(const void *vol, int vol_dtype, int VX, int VY, int VZ,
                      int64_t sx, int64_t sy, int64_t sz, int64_t vol_view_stride,
                      const float *tf, int R, int64_t tf_view_stride,
                      const float *cam, const float *entry, const float *exit_, const float *rays,
                      const int32_t *nsamp, int n_views, int W, int H, int max_samples,
                      float sampling_rate, double fov_rad, double near_plane, int variant,
                      const float *grad_out, const float *out_rgba,
                      float *d_vol, int64_t dsx, int64_t dsy, int64_t dsz, int64_t dvol_view_stride,
                      float *d_tf, int64_t dtf_view_stride,
                      void *workspace, size_t workspace_bytes, int img_W, int row0, void *stream);

/* The one exchange step of the path when views (or image bands) are sharded over the GPUs of a node: an in-place float32
 * SUM all-reduce of the shared gradients over RCCL / xGMI (the reference is single-device and has no
 * counterpart). For hosts without torch.distributed -- the Python package itself uses torch's "nccl" backend, which is the
 * same RCCL. RCCL is dlopen()ed on first use (DR_EUNSUPPORTED if absent); a communicator is an opaque ncclComm_t.
 *   one process per GPU:   rank 0 calls dr_comm_unique_id(id) and ships the 128 bytes to its peers by any means;
 *                          every rank: hipSetDevice(its GPU); dr_comm_init_rank(&comm, n_ranks, id, rank)
 *   one process, n GPUs:   dr_comm_init_all(comms, n, devices) (devices NULL = 0..n-1), then one thread (or a group) per device
 *   per step:              dr_allreduce_gradients_f32(comm, d_vol, n_vol, d_tf, n_tf, stream)   (either may be NULL)
 * d_volume must be ONE dense block of n_vol floats (any axis order: the sum is elementwise), as dr_march_bwd fills it when
 * its strides describe a permutation of a contiguous tensor. The calls are asynchronous on `stream`. */
int dr_comm_unique_id(void *id128);
int dr_comm_init_rank(void **comm, int n_ranks, const void *id128, int rank);
int dr_comm_init_all(void **comms, int n_devices, const int *devices);
int dr_allreduce_f32(void *comm, float *buf, size_t n, void *stream);
int dr_allreduce_gradients_f32(void *comm, float *d_vol, size_t n_vol, float *d_tf, size_t n_tf, void *stream);
int dr_comm_destroy(void *comm);

/* Image loss and its gradient, one pass over the rendered image (the step after the march in an optimisation
 * loop): replaces compute_loss (examples/taichi_volume_raycaster.py:368-373, "EX.py") and the torch mse_loss
 * round trip of EX.py:439-443.
 *   out_rgba, reference  [n] f32 (n = n_views*W*H*4)
 *   grad_out [n] f32, nullable:  (out - ref) * (2*inv_norm)
 *   loss     one f64 on the device, nullable; ACCUMULATED into:  += inv_norm * sum (out - ref)^2
 * inv_norm = 1/n gives torch.nn.functional.mse_loss; 1/(3*W*H) gives EX.py:369-373. */
int dr_mse_loss_grad(const float *out_rgba, const float *reference, int64_t n, float inv_norm,
                     float *grad_out, double *loss, void *stream);

/* Momentum gradient step on the transfer function, in place (apply_grad, EX.py:375-381):
 *   momentum = gamma*momentum + lr*clamp(d_tf, -max_grad, max_grad);  tf = max(tf - momentum, 0)
 *   tf, d_tf, momentum [n] f32 (n = R*4). */
int dr_tf_momentum_step(float *tf, const float *d_tf, float *momentum, int n, float lr, float gamma,
                        float max_grad, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DIFFERENDER_HIP_H */
