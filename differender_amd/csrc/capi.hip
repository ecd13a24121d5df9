// capi.hip -- extern "C" boundary of libdifferender_hip.so (declared in include/differender_hip.h).
// Argument validation + dispatch only; kernels live in the other translation units.
#include <hip/hip_runtime.h>

#include "../../include/differender_hip.h"
#include "dr_kernels.h"

#include <mutex>
#include <unordered_map>

using namespace dr;

namespace dr {
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device, size) instead of on every launch
hipError_t allow_lds_impl(const void *kernel, size_t bytes) {
    static std::mutex mu;
    static std::unordered_map<const void *, std::unordered_map<int, size_t>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    size_t &have = done[kernel][dev];
    if (have >= bytes) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) have = bytes;
    return e;
}
}  // namespace dr

namespace {
// The library runs on the device that owns the caller's buffers, whatever the calling thread's current device is
// (autograd calls backward from another thread than forward): switch for the duration of the call, restore afterwards.
struct DeviceOf {
    int prev = -1, dev = -1;
    hipError_t err = hipSuccess;
    explicit DeviceOf(const void *device_ptr) {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, device_ptr) != hipSuccess) { (void)hipGetLastError(); return; }  // not a device pointer we know: leave as is
        dev = attr.device;
        // (a launch error somebody else left behind -- an earlier failed call of ours, another library -- must not be taken for
        // this call's: the launch functions read hipGetLastError() after their launches)
        (void)hipGetLastError();
        if (hipGetDevice(&prev) != hipSuccess) { prev = -1; return; }
        if (prev != dev) err = hipSetDevice(dev); else prev = -1;
    }
    ~DeviceOf() { if (prev >= 0) (void)hipSetDevice(prev); }
};
}  // namespace

extern "C" {

// OR-ed at load time by every translation unit that was compiled with a what-if or diagnostic switch (dr_experiment.h)
int dr_experiment_flags_ = 0;

int dr_build_flags(void) { return dr_experiment_flags_; }
// a library that computes wrong results on purpose does not answer with the ABI version a loader expects
int dr_abi_version(void) { return (dr_experiment_flags_ & DR_BUILD_WRONG_RESULTS) ? -DR_ABI_VERSION : DR_ABI_VERSION; }

const char *dr_error_string(int code) {
    switch (code) {
        case 0: return "success";
        case DR_EINVAL: return "differender_hip: invalid argument";
        case DR_EUNSUPPORTED: return "differender_hip: unsupported configuration";
        case DR_ECOLLECTIVE: return "differender_hip: RCCL reported an error";
        default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "differender_hip: unknown error";
}

size_t dr_workspace_bytes(int n_views, int W, int H, int VX, int VY, int VZ, int R) {
    if (n_views <= 0 || W <= 0 || H <= 0 || VX < 2 || VY < 2 || VZ < 2 || R < 1) return 0;
    if (!brick_path_supported(VX, VY, VZ, R) || !brick_image_supported(W, H, VX, VY, VZ)) return 0;
    return brick_workspace_bytes(n_views, W, H, VX, VY, VZ);
}

size_t dr_workspace_bytes_tape(int n_views, int W, int H, int VX, int VY, int VZ, int R, int max_samples, float sampling_rate) {
    if (dr_workspace_bytes(n_views, W, H, VX, VY, VZ, R) == 0 || max_samples < 0 || !(sampling_rate > 0.0f)) return 0;
    return brick_workspace_bytes_tape(n_views, W, H, VX, VY, VZ, max_samples, sampling_rate);
}

int dr_ray_setup_rows(const float *cam, int n_views, int W, int H, int img_W, int row0, int VX, int VY, int VZ,
                      double fov_rad, double near_plane, float sampling_rate, uint32_t jitter_seed, uint32_t view_base,
                      float *entry, float *exit_, float *rays, int32_t *nsamp, void *stream) {
    if (!cam || !entry || !exit_ || !rays || !nsamp) return DR_EINVAL;
    if (n_views <= 0 || W <= 0 || H <= 0 || VX < 2 || VY < 2 || VZ < 2) return DR_EINVAL;
    if (n_views > 65535 || !(sampling_rate > 0.0f)) return DR_EINVAL;
    if (row0 < 0 || img_W < W || row0 > img_W - W) return DR_EINVAL;
    DeviceOf guard(entry);
    if (guard.err != hipSuccess) return (int)guard.err;
    return (int)launch_ray_setup(cam, n_views, W, H, img_W, row0, VX, VY, VZ, fov_rad, near_plane, sampling_rate,
                                 jitter_seed, view_base, entry, exit_, rays, nsamp, (hipStream_t)stream);
}

int dr_ray_setup(const float *cam, int n_views, int W, int H, int VX, int VY, int VZ, double fov_rad,
                 double near_plane, float sampling_rate, uint32_t jitter_seed, uint32_t view_base, float *entry,
                 float *exit_, float *rays, int32_t *nsamp, void *stream) {
    return dr_ray_setup_rows(cam, n_views, W, H, W, 0, VX, VY, VZ, fov_rad, near_plane, sampling_rate, jitter_seed,
                             view_base, entry, exit_, rays, nsamp, stream);
}

static int fill_common(MarchArgs &a, const void *vol, int vol_dtype, int VX, int VY, int VZ, int64_t sx, int64_t sy,
                       int64_t sz, int64_t vol_view_stride, const float *tf, int R, int64_t tf_view_stride,
                       const float *cam, const float *entry, const float *exit_, const float *rays,
                       const int32_t *nsamp, int n_views, int W, int H, int max_samples, float sampling_rate) {
    if (!vol || !tf || !cam || !entry || !exit_ || !rays || !nsamp) return DR_EINVAL;
    if (vol_dtype != DR_F32 && vol_dtype != DR_F16) return DR_EINVAL;
    if (n_views <= 0 || n_views > 65535 || W <= 0 || H <= 0 || VX < 2 || VY < 2 || VZ < 2 || R < 1) return DR_EINVAL;
    if (max_samples < 0 || !(sampling_rate > 0.0f)) return DR_EINVAL;
    if (tf_view_stride % 4 != 0) return DR_EINVAL;
    a = MarchArgs{};
    a.vol = vol; a.vol_dtype = vol_dtype; a.VX = VX; a.VY = VY; a.VZ = VZ;
    a.sx = sx; a.sy = sy; a.sz = sz; a.vol_vs = vol_view_stride;
    a.tf = tf; a.R = R; a.tf_vs = tf_view_stride;
    a.cam = cam; a.entry = entry; a.exit_ = exit_; a.rays = rays; a.nsamp = nsamp;
    a.n_views = n_views; a.W = W; a.H = H; a.S = max_samples; a.sr = sampling_rate;
    return 0;
}

int dr_march_fwd_rows(const void *vol, int vol_dtype, int VX, int VY, int VZ, int64_t sx, int64_t sy, int64_t sz,
                 int64_t vol_view_stride, const float *tf, int R, int64_t tf_view_stride, const float *cam,
                 const float *entry, const float *exit_, const float *rays, const int32_t *nsamp, int n_views, int W,
                 int H, int max_samples, float sampling_rate, double fov_rad, double near_plane, int mode, int variant,
                 float *out_rgba, int32_t *steps, void *workspace, size_t workspace_bytes, int img_W, int row0,
                      void *stream) {
    MarchArgs a;
    int rc = fill_common(a, vol, vol_dtype, VX, VY, VZ, sx, sy, sz, vol_view_stride, tf, R, tf_view_stride, cam,
                         entry, exit_, rays, nsamp, n_views, W, H, max_samples, sampling_rate);
    if (rc) return rc;
    if (!out_rgba) return DR_EINVAL;
    if (row0 < 0 || img_W < W || row0 > img_W - W) return DR_EINVAL;
    a.img_W = img_W; a.row0 = row0;
    if (mode != DR_MODE_DIFF && mode != DR_MODE_NONDIFF) return DR_EINVAL;
    const int hints = variant & ~0xff;
    variant &= 0xff;
    if (variant < DR_VARIANT_AUTO || variant > DR_VARIANT_BASELINE) return DR_EINVAL;
    if (hints & ~(DR_HINT_NO_EARLY_TERMINATION | DR_HINT_EARLY_TERMINATION | DR_COUNT_EVALUATED | DR_TAPE_TF)) return DR_EINVAL;
    if ((hints & DR_TAPE_TF) && mode != DR_MODE_DIFF) return DR_EINVAL;
    if ((hints & DR_HINT_NO_EARLY_TERMINATION) && (hints & DR_HINT_EARLY_TERMINATION)) return DR_EINVAL;
    DeviceOf guard(vol);
    if (guard.err != hipSuccess) return (int)guard.err;
    a.mode = mode; a.out = out_rgba; a.steps = steps; a.hints = hints;
    a.fov_rad = fov_rad; a.near_plane = near_plane; a.workspace = workspace; a.workspace_bytes = workspace_bytes;
    if (variant != DR_VARIANT_BASELINE && workspace && brick_path_supported(VX, VY, VZ, R) &&
        brick_image_supported(W, H, VX, VY, VZ)) {
        if (workspace_bytes < ((hints & DR_TAPE_TF) ? brick_workspace_bytes_tape(n_views, W, H, VX, VY, VZ, max_samples, sampling_rate)
                                                    : brick_workspace_bytes(n_views, W, H, VX, VY, VZ))) return DR_EINVAL;
        if (flat_strides_ok(sx, sy, sz)) return launch_march_fwd_flat(a, (hipStream_t)stream);
    }
    // served by the plain kernels: whatever coarse tape the workspace still holds is not this call's
    if (workspace) {
        const hipError_t e = flat_invalidate_workspace(workspace, workspace_bytes, (hipStream_t)stream);
        if (e != hipSuccess) return (int)e;
    }
    return launch_march_fwd_baseline(a, (hipStream_t)stream);
}

int dr_march_fwd(const void *vol, int vol_dtype, int VX, int VY, int VZ, int64_t sx, int64_t sy, int64_t sz,
                 int64_t vol_view_stride, const float *tf, int R, int64_t tf_view_stride, const float *cam,
                 const float *entry, const float *exit_, const float *rays, const int32_t *nsamp, int n_views, int W,
                 int H, int max_samples, float sampling_rate, double fov_rad, double near_plane, int mode, int variant,
                 float *out_rgba, int32_t *steps, void *workspace, size_t workspace_bytes, void *stream) {
    return dr_march_fwd_rows(vol, vol_dtype, VX, VY, VZ, sx, sy, sz, vol_view_stride, tf, R, tf_view_stride, cam, entry,
                             exit_, rays, nsamp, n_views, W, H, max_samples, sampling_rate, fov_rad, near_plane, mode,
                             variant, out_rgba, steps, workspace, workspace_bytes, W, 0, stream);
}

int dr_march_bwd_variant(int n_views, int W, int H, int VX, int VY, int VZ, int R, int64_t sx, int64_t sy, int64_t sz,
                         int64_t dsx, int64_t dsy, int64_t dsz, int has_dvol, int variant, int has_workspace) {
    const bool fast = variant != DR_VARIANT_BASELINE && has_workspace && n_views > 0 && W > 0 && H > 0 && VX >= 2 &&
                      VY >= 2 && VZ >= 2 && R >= 1 && brick_path_supported(VX, VY, VZ, R) &&
                      brick_image_supported(W, H, VX, VY, VZ) && flat_strides_ok(sx, sy, sz) &&
                      (!has_dvol || flat_strides_ok(dsx, dsy, dsz));
    return fast ? DR_VARIANT_AUTO : DR_VARIANT_BASELINE;
}

int dr_march_bwd_rows(const void *vol, int vol_dtype, int VX, int VY, int VZ, int64_t sx, int64_t sy, int64_t sz,
                 int64_t vol_view_stride, const float *tf, int R, int64_t tf_view_stride, const float *cam,
                 const float *entry, const float *exit_, const float *rays, const int32_t *nsamp, int n_views, int W,
                 int H, int max_samples, float sampling_rate, double fov_rad, double near_plane, int variant,
                 const float *grad_out, const float *out_rgba, float *d_vol, int64_t dsx, int64_t dsy, int64_t dsz,
                 int64_t dvol_view_stride, float *d_tf, int64_t dtf_view_stride, void *workspace,
                 size_t workspace_bytes, int img_W, int row0, void *stream) {
    MarchArgs a;
    int rc = fill_common(a, vol, vol_dtype, VX, VY, VZ, sx, sy, sz, vol_view_stride, tf, R, tf_view_stride, cam,
                         entry, exit_, rays, nsamp, n_views, W, H, max_samples, sampling_rate);
    if (rc) return rc;
    if (!grad_out || !out_rgba) return DR_EINVAL;
    if (row0 < 0 || img_W < W || row0 > img_W - W) return DR_EINVAL;
    a.img_W = img_W; a.row0 = row0;
    const int bwd_flags = variant & ~0xff;
    variant &= 0xff;
    if (bwd_flags & ~(DR_COUNT_EVALUATED | DR_TAPE_TF)) return DR_EINVAL;
    if ((bwd_flags & DR_TAPE_TF) && d_vol) return DR_EINVAL;   // the tape serves the TF-only backward
    if (variant < DR_VARIANT_AUTO || variant > DR_VARIANT_BASELINE) return DR_EINVAL;
    if (dtf_view_stride % 4 != 0) return DR_EINVAL;
    if (!d_vol && !d_tf) return 0;  // nothing requested
    DeviceOf guard(vol);
    if (guard.err != hipSuccess) return (int)guard.err;
    a.mode = DR_MODE_DIFF; a.hints = bwd_flags;
    a.grad_out = grad_out; a.out_fwd = out_rgba;
    a.d_vol = d_vol; a.dsx = dsx; a.dsy = dsy; a.dsz = dsz; a.dvol_vs = dvol_view_stride;
    a.d_tf = d_tf; a.dtf_vs = dtf_view_stride;
    a.fov_rad = fov_rad; a.near_plane = near_plane; a.workspace = workspace; a.workspace_bytes = workspace_bytes;
    if (dr_march_bwd_variant(n_views, W, H, VX, VY, VZ, R, sx, sy, sz, dsx, dsy, dsz, d_vol != nullptr, variant,
                             workspace != nullptr) == DR_VARIANT_AUTO) {
        if (workspace_bytes < ((bwd_flags & DR_TAPE_TF) ? brick_workspace_bytes_tape(n_views, W, H, VX, VY, VZ, max_samples, sampling_rate)
                                                        : brick_workspace_bytes(n_views, W, H, VX, VY, VZ))) return DR_EINVAL;
        return launch_march_bwd_flat(a, (hipStream_t)stream);
    }
    return launch_march_bwd_baseline(a, (hipStream_t)stream);
}

int dr_march_bwd(const void *vol, int vol_dtype, int VX, int VY, int VZ, int64_t sx, int64_t sy, int64_t sz,
                 int64_t vol_view_stride, const float *tf, int R, int64_t tf_view_stride, const float *cam,
                 const float *entry, const float *exit_, const float *rays, const int32_t *nsamp, int n_views, int W,
                 int H, int max_samples, float sampling_rate, double fov_rad, double near_plane, int variant,
                 const float *grad_out, const float *out_rgba, float *d_vol, int64_t dsx, int64_t dsy, int64_t dsz,
                 int64_t dvol_view_stride, float *d_tf, int64_t dtf_view_stride, void *workspace,
                 size_t workspace_bytes, void *stream) {
    return dr_march_bwd_rows(vol, vol_dtype, VX, VY, VZ, sx, sy, sz, vol_view_stride, tf, R, tf_view_stride, cam, entry,
                             exit_, rays, nsamp, n_views, W, H, max_samples, sampling_rate, fov_rad, near_plane, variant,
                             grad_out, out_rgba, d_vol, dsx, dsy, dsz, dvol_view_stride, d_tf, dtf_view_stride,
                             workspace, workspace_bytes, W, 0, stream);
}

int dr_mse_loss_grad(const float *out_rgba, const float *reference, int64_t n, float inv_norm, float *grad_out,
                     double *loss, void *stream) {
    if (!out_rgba || !reference || n <= 0 || (!grad_out && !loss)) return DR_EINVAL;
    DeviceOf guard(out_rgba);
    if (guard.err != hipSuccess) return (int)guard.err;
    return (int)launch_mse_loss_grad(out_rgba, reference, n, inv_norm, grad_out, loss, (hipStream_t)stream);
}

int dr_tf_momentum_step(float *tf, const float *d_tf, float *momentum, int n, float lr, float gamma, float max_grad,
                        void *stream) {
    if (!tf || !d_tf || !momentum || n <= 0 || !(max_grad >= 0.0f)) return DR_EINVAL;
    DeviceOf guard(tf);
    if (guard.err != hipSuccess) return (int)guard.err;
    return (int)launch_tf_momentum_step(tf, d_tf, momentum, n, lr, gamma, max_grad, (hipStream_t)stream);
}

}  // extern "C"
