// epilogue.hip -- the two small kernels either side of the march in an optimisation loop:
//   * image loss + its gradient (EX.py:368-373 compute_loss / the torch mse_loss of EX.py:439-443),
//   * momentum step of the transfer function (EX.py:375-381 apply_grad).
// Both are elementwise over <= a few MiB; they exist so that one iteration (render, loss, backward, update)
// needs no host round trip and no framework elementwise launches between the march kernels.
#include <hip/hip_runtime.h>

#include "dr_kernels.h"

namespace dr {

// grad = (out - ref) * two_inv_norm;  *loss += inv_norm * sum (out - ref)^2   (sum carried in double)
__global__ __launch_bounds__(256) void mse_loss_grad_kernel(const float *__restrict__ out,
                                                            const float *__restrict__ ref, int64_t n,
                                                            float inv_norm, float two_inv_norm,
                                                            float *__restrict__ grad, double *__restrict__ loss) {
    __shared__ double part[4];
    double acc = 0.0;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float d = out[i] - ref[i];
        if (grad) grad[i] = d * two_inv_norm;
        acc += (double)d * (double)d;
    }
    if (!loss) return;
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) part[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = (part[0] + part[1]) + (part[2] + part[3]);
        atomicAdd(loss, s * (double)inv_norm);
    }
}

// m = gamma*m + lr*clamp(g, -c, c);  tf = max(tf - m, 0)
__global__ __launch_bounds__(256) void tf_momentum_step_kernel(float *__restrict__ tf, const float *__restrict__ g,
                                                               float *__restrict__ mom, int n, float lr, float gamma,
                                                               float max_grad) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float c = fminf(max_grad, fmaxf(-max_grad, g[i]));
    float m = gamma * mom[i] + lr * c;
    mom[i] = m;
    tf[i] = fmaxf(tf[i] - m, 0.0f);
}

hipError_t launch_mse_loss_grad(const float *out, const float *ref, int64_t n, float inv_norm, float *grad,
                                double *loss, hipStream_t stream) {
    int64_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;  // 4 blocks per CU; each thread then streams n/262144 elements
    hipLaunchKernelGGL(mse_loss_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, out, ref, n, inv_norm,
                       2.0f * inv_norm, grad, loss);
    return hipGetLastError();
}

hipError_t launch_tf_momentum_step(float *tf, const float *g, float *mom, int n, float lr, float gamma,
                                   float max_grad, hipStream_t stream) {
    hipLaunchKernelGGL(tf_momentum_step_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, tf, g, mom, n, lr,
                       gamma, max_grad);
    return hipGetLastError();
}

}  // namespace dr
