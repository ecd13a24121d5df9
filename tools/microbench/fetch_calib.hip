// fetch_calib.hip -- calibrates rocprofv3's FETCH_SIZE for the access pattern of the brick staging (MI355X guide:
// "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
// Kernel A streams a 512^3 f32 volume with 16-B loads (known: 512 MiB). Kernel B reads it the way the march
// kernels stage their bricks: one workgroup per 12^3 brick loads its 15^3 box in rows of 15 floats, 4 B per lane
// (known: requested bytes, unique bytes = 512 MiB). Run under rocprofv3 --pmc FETCH_SIZE and compare.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int N = 512, BRK = 12, BOX = 15, NB = (N - 1 + BRK - 1) / BRK;

__global__ __launch_bounds__(256) void stream_read(const float4 *p, size_t n4, float *out) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(256) void box_read(const float *vol, float *out, unsigned long long *req) {
    const int b = blockIdx.x;
    const int bz = b % NB, by = (b / NB) % NB, bx = b / (NB * NB);
    const int ox = bx * BRK - 1, oy = by * BRK - 1, oz = bz * BRK - 1;
    const int a = threadIdx.x & 15, row = threadIdx.x >> 4;
    float acc = 0.f;
    int cnt = 0;
    for (int r = row; r < BOX * BOX; r += 16) {
        const int d = r / BOX, bb = r % BOX;  // (x, y) of the row; a walks z (contiguous)
        const int x = ox + d, y = oy + bb, z = oz + a;
        if (a < BOX && (unsigned)x < (unsigned)N && (unsigned)y < (unsigned)N && (unsigned)z < (unsigned)N) {
            acc += vol[((size_t)x * N + y) * N + z];
            ++cnt;
        }
    }
    if (acc == 12345.678f) out[0] = acc;
    atomicAdd(req, (unsigned long long)cnt * 4ull);
}
int main() {
    const size_t n = (size_t)N * N * N;
    float *vol, *out; unsigned long long *req;
    CK(hipMalloc(&vol, n * 4)); CK(hipMalloc(&out, 4)); CK(hipMalloc(&req, 8));
    CK(hipMemset(vol, 0, n * 4)); CK(hipMemset(req, 0, 8));
    // a 1 GiB spoiler between the two so that neither finds the other's lines in the 256 MiB infinity cache
    float *spoil; CK(hipMalloc(&spoil, (size_t)1 << 30)); CK(hipMemset(spoil, 1, (size_t)1 << 30));
    hipLaunchKernelGGL(stream_read, dim3(4096), dim3(256), 0, 0, reinterpret_cast<const float4 *>(vol), n / 4, out);
    CK(hipDeviceSynchronize());
    CK(hipMemset(spoil, 2, (size_t)1 << 30));
    hipLaunchKernelGGL(box_read, dim3(NB * NB * NB), dim3(256), 0, 0, vol, out, req);
    CK(hipDeviceSynchronize());
    unsigned long long h; CK(hipMemcpy(&h, req, 8, hipMemcpyDeviceToHost));
    printf("stream_read: known bytes %zu\nbox_read: requested bytes %llu, unique bytes %zu\n", n * 4, h, n * 4);
    return 0;
}
