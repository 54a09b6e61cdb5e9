// mfma_ceiling.hip - what a K loop of v_mfma_f32_32x32x16_f16 can reach on this chip, measured (not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -o build_dbg/mfma_ceiling tools/mfma_ceiling.hip && build_dbg/mfma_ceiling
// Arms (8 waves per workgroup = 2 per SIMD, one workgroup per CU, 64x64 accumulators per wave = the conv kernel's shape):
//   regs      24 MFMAs per step on operands held in registers                     -> the matrix pipe + clock under load
//   lds       + the conv loop's 16 ds_read_b128 per step, read-then-wait           -> + LDS operand delivery
//   lds_pipe  the same reads issued one half-step ahead of their MFMAs             -> + software pipelining
//   lds_bar   lds + one __syncthreads per step                                     -> + the per-step barrier
// each with zero and with random operand data (DVFS: the clock under load depends on the toggling).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(512, 1) void k_loop(const f16x8* __restrict__ src, float* __restrict__ out, int steps) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // fill 96 KB of LDS from src (random or zero)
    for (int i = tid; i < 96 * 1024 / 16; i += 512) reinterpret_cast<f16x8*>(lds)[i] = src[(blockIdx.x * 6144 + i) & 0xffff];
    __syncthreads();
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    // register operands for MODE 0
    f16x8 ra[2][2][2], rb[2][2][2];      // [mi|ni][plane][khalf]
    for (int a = 0; a < 2; ++a)
        for (int pl = 0; pl < 2; ++pl)
            for (int h = 0; h < 2; ++h) {
                ra[a][pl][h] = src[(tid * 8 + a * 4 + pl * 2 + h) & 0xffff];
                rb[a][pl][h] = src[(tid * 8 + a * 4 + pl * 2 + h + 4096) & 0xffff];
            }
    const char* abase = lds + (wave >> 1) * 8192 + (lane & 31) * 128;
    const char* bbase = lds + 49152 + (wave & 1) * 8192 + (lane & 31) * 128;
    auto frag = [&](const char* base, int t, int pl, int h, int stage) {
        // rows of 128 B: plane pl in bytes [64 pl, 64 pl + 64), k-half h -> +32, XOR swizzle by row
        const int row = lane & 31, key = ((row >> 1) & 7) ^ ((row & 1) << 2);
        return *reinterpret_cast<const f16x8*>(base + stage * 8192 + t * 4096 + (((pl * 4 + h * 2 + (lane >> 5)) ^ key) << 4));
    };
    auto mm = [&](f16x8 (&fa)[2][2], f16x8 (&fb)[2][2]) {    // [tile][plane] of one k-half: 12 MFMAs
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int pa = (t == 2) ? 1 : 0, pb = (t == 1) ? 1 : 0;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mi][pa], fb[ni][pb], acc[mi][ni], 0, 0, 0);
        }
    };
    if (MODE == 0) {
        for (int s = 0; s < steps; ++s) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f16x8 fa[2][2], fb[2][2];
                for (int a = 0; a < 2; ++a)
                    for (int pl = 0; pl < 2; ++pl) { fa[a][pl] = ra[a][pl][h]; fb[a][pl] = rb[a][pl][h]; }
                mm(fa, fb);
            }
        }
    } else if (MODE == 1 || MODE == 3) {
        for (int s = 0; s < steps; ++s) {
            const int stage = s % 3;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f16x8 fa[2][2], fb[2][2];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) {
                        fa[a][pl] = frag(abase, a, pl, h, stage);
                        fb[a][pl] = frag(bbase, a, pl, h, stage);
                    }
                mm(fa, fb);
            }
            if (MODE == 3) __syncthreads();
        }
    } else {
        f16x8 fa[2][2][2], fb[2][2][2];      // [set][tile][plane]
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) { fa[0][a][pl] = frag(abase, a, pl, 0, 0); fb[0][a][pl] = frag(bbase, a, pl, 0, 0); }
        for (int s = 0; s < steps; ++s) {
            const int stage = s % 3, nstage = (s + 1) % 3;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) { fa[1][a][pl] = frag(abase, a, pl, 1, stage); fb[1][a][pl] = frag(bbase, a, pl, 1, stage); }
            mm(fa[0], fb[0]);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) { fa[0][a][pl] = frag(abase, a, pl, 0, nstage); fb[0][a][pl] = frag(bbase, a, pl, 0, nstage); }
            mm(fa[1], fb[1]);
        }
    }
    float sum = 0.f;
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
            for (int r = 0; r < 16; ++r) sum += acc[a][b][r];
    out[blockIdx.x * 512 + tid] = sum;
}

template <int MODE>
static double run(const char* name, const f16x8* src, float* out, int blocks, int steps, const char* data) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_loop<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_loop<MODE>, dim3(blocks), dim3(512), 96 * 1024, 0, src, out, steps);
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_loop<MODE>, dim3(blocks), dim3(512), 96 * 1024, 0, src, out, steps);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    // per step per wave 24 MFMAs of 32x32x16: 2*32*32*16 flops each; "useful" fp32-equivalent flops = 1/3 of them
    const double mf = (double)blocks * 8 * steps * 24.0 * 2 * 32 * 32 * 16;
    const double tf = mf / (ms * 1e-3) / 1e12;
    printf("%-9s %-6s blocks %5d steps %5d  %8.3f ms  %7.1f TF fp16-MFMA  = %6.1f TF fp32-equivalent (3 products)  = %4.1f %% of 2500\n",
           name, data, blocks, steps, ms, tf, tf / 3, 100 * tf / 2500);
    fflush(stdout);
    return tf;
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 2000;
    std::vector<_Float16> h(65536 * 8);
    f16x8* src;
    float* out;
    CK(hipMalloc(&src, h.size() * 2));
    CK(hipMalloc(&out, 4096 * 512 * 4));
    for (int data = 0; data < 2; ++data) {
        srand(1);
        for (auto& v : h) v = data ? (_Float16)((rand() % 2001 - 1000) * 1e-3f) : (_Float16)0.f;
        CK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        const char* dn = data ? "random" : "zero";
        for (int blocks : {256, 676}) {
            const int st = blocks == 256 ? steps : steps / 2;
            run<0>("regs", src, out, blocks, st, dn);
            run<1>("lds", src, out, blocks, st, dn);
            run<2>("lds_pipe", src, out, blocks, st, dn);
            run<3>("lds_bar", src, out, blocks, st, dn);
        }
    }
    return 0;
}
