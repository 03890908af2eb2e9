// Does the time-major layout ([step][cell], one row = every cell) cost the marching kernels anything?  Each thread marches one cell
// through T steps and reads / writes NS streams, one dword per stream and step, with the vertical kernels' addressing (wave-uniform
// row + lane offset) in two layouts: rows of `npad` cells (consecutive steps of a cell are npad*4 bytes apart: a new page per stream
// and step) and blocks of 256 cells (consecutive steps of a workgroup are 1 KB apart).  Same bytes, same instruction stream.
// hipcc -O3 --offload-arch=gfx950 -o march_probe march_probe.hip ; ./march_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int AUX>
__device__ __forceinline__ float row_load(const float* row, unsigned off) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)row, 0, 0x7fffffff, 0x00020000);
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, AUX));
}
template <int AUX>
__device__ __forceinline__ void row_store(float* row, unsigned off, float v) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)row, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, off, 0, AUX);
}

// MODE 0: NS load streams; MODE 1: one load stream + NS store streams.  stride = floats between consecutive steps of a row.
template <int NS, int MODE, int WORK>
__global__ __launch_bounds__(256) void march(float* const* bufs, float* out, int T, size_t stride, size_t blockbase) {
    const size_t base = (size_t)blockIdx.x * blockbase;          // start of this workgroup's rows
    const unsigned off = threadIdx.x * 4u;
    float acc = 0.f, nx[NS];
    const float* in[NS]; float* o[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) { in[s] = bufs[s] + base; o[s] = bufs[s] + base; }
    const float* in0 = bufs[NS] + base;
    if (MODE == 0) {
#pragma unroll
        for (int s = 0; s < NS; ++s) { nx[s] = row_load<0>(in[s], off); in[s] += stride; }
    } else { nx[0] = row_load<0>(in0, off); in0 += stride; }
    for (int t = 0; t < T; ++t) {
        float cur[NS];
        if (MODE == 0) {
#pragma unroll
            for (int s = 0; s < NS; ++s) { cur[s] = nx[s]; asm volatile("" : "+v"(cur[s])); }
            if (t + 1 < T) {
#pragma unroll
                for (int s = 0; s < NS; ++s) { nx[s] = row_load<0>(in[s], off); in[s] += stride; }
            }
        } else {
            cur[0] = nx[0]; asm volatile("" : "+v"(cur[0]));
            if (t + 1 < T) { nx[0] = row_load<0>(in0, off); in0 += stride; }
#pragma unroll
            for (int s = 1; s < NS; ++s) cur[s] = cur[0] + (float)s;
        }
        float x = cur[0];
#pragma unroll
        for (int s = 1; s < NS; ++s) x += cur[s];
#pragma unroll 8
        for (int w = 0; w < WORK; ++w) x = fmaf(x, 0.999f, acc * 1e-3f + 0.5f);
        acc += x;
        if (MODE == 1) {
#pragma unroll
            for (int s = 0; s < NS; ++s) { row_store<0>(o[s], off, x + (float)s); o[s] += stride; }
        }
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int NS, int MODE, int WORK>
static double run(float* const* dbufs, float* dout, int nblk, int T, bool blocked) {
    const size_t npad = (size_t)nblk * 256;
    const size_t stride = blocked ? 256 : npad, blockbase = blocked ? (size_t)T * 256 : 256;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((march<NS, MODE, WORK>), dim3(nblk), dim3(256), 0, 0, dbufs, dout, T, stride, blockbase);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    return best;
}

int main(int argc, char** argv) {
    const int nblk = 4096, T = argc > 1 ? atoi(argv[1]) : 2048;
    const size_t n = (size_t)nblk * 256 * T;
    constexpr int NB = 5;
    std::vector<float*> h(NB);
    for (int i = 0; i < NB; ++i) { CK(hipMalloc((void**)&h[i], n * 4)); CK(hipMemset(h[i], 0, n * 4)); }
    float** dbufs; CK(hipMalloc((void**)&dbufs, NB * sizeof(float*))); CK(hipMemcpy(dbufs, h.data(), NB * sizeof(float*), hipMemcpyHostToDevice));
    float* dout; CK(hipMalloc((void**)&dout, (size_t)nblk * 256 * 4));
    const double gb4 = 4.0 * n * 4 / 1e9;
    printf("{\"cells\": %d, \"steps\": %d", nblk * 256, T);
    for (int blocked = 0; blocked < 2; ++blocked) {
        const char* L = blocked ? "blocked256" : "rows";
        double ms;
        ms = run<4, 0, 8>(dbufs, dout, nblk, T, blocked);   printf(", \"load4_work8_%s_ms\": %.2f, \"load4_work8_%s_TBps\": %.2f", L, ms, L, gb4 / ms);
        ms = run<4, 0, 160>(dbufs, dout, nblk, T, blocked); printf(", \"load4_work160_%s_ms\": %.2f", L, ms);
        ms = run<4, 1, 8>(dbufs, dout, nblk, T, blocked);   printf(", \"store4_work8_%s_ms\": %.2f, \"store4_work8_%s_TBps\": %.2f", L, ms, L, gb4 / ms);
        ms = run<4, 1, 80>(dbufs, dout, nblk, T, blocked);  printf(", \"store4_work80_%s_ms\": %.2f", L, ms);
    }
    printf("}\n");
    return 0;
}
