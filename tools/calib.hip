// tools/calib.hip -- calibration micro-kernels for the numbers bench.py's roofline blocks are priced against (gfx950).
//
//   tools/_build/calib hbm    known-byte streaming kernels in the access widths this path uses, to be run under
//                             `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes):
//                             the ratio known / counted per kernel is the correction factor for that width
//                             (MI355X_MICROARCH.md: FETCH_SIZE is only calibrated for 16 B/lane reads: x2).
//   tools/_build/calib valu   issue cost (shader cycles per wave64 instruction per SIMD) of the vector-ALU instruction
//                             classes k_resolve / k_normals_interior are made of, at 1 and at 8 waves per SIMD, from
//                             s_memtime around long independent instruction streams.  Prints one JSON line.
//
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/_build/calib tools/calib.hip      (tools/build_calib.sh)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)

// ---- HBM counter calibration ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void cal_read(const T* __restrict__ in, size_t n, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const T v = in[i];
        const uint32_t* w = reinterpret_cast<const uint32_t*>(&v);
#pragma unroll
        for (unsigned k = 0; k < sizeof(T) / 4; ++k) acc ^= w[k];
    }
    if (acc == 0x12345678u) sink[0] = acc;      // never true for the test pattern; keeps the loads alive
}
// k_resolve's read pattern: a (16+2) x (64+2) tile of 8-byte keys per workgroup out of a W x H image, clamp-to-edge
__global__ __launch_bounds__(256) void cal_read8_tile(const uint64_t* __restrict__ in, int W, int H, uint32_t* __restrict__ sink) {
    const int bx = blockIdx.x * 64, by = blockIdx.y * 16;
    uint32_t acc = 0;
    for (int idx = threadIdx.x; idx < 18 * 66; idx += 256) {
        const int ly = idx / 66, lx = idx - ly * 66;
        int x = bx + lx - 1, y = by + ly - 1;
        x = x < 0 ? 0 : (x > W - 1 ? W - 1 : x);
        y = y < 0 ? 0 : (y > H - 1 ? H - 1 : y);
        const uint64_t v = in[(size_t)blockIdx.z * W * H + (size_t)y * W + x];
        acc ^= (uint32_t)v ^ (uint32_t)(v >> 32);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
template <typename T>
__global__ __launch_bounds__(256) void cal_write(T* __restrict__ out, size_t n, T v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = v;
}
// the raster kernels' sink: one blind 64-bit atomic min per lane, lanes of a wave on consecutive keys
__global__ __launch_bounds__(256) void cal_atomic8(uint64_t* __restrict__ out, size_t n, uint64_t v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        atomicMin(reinterpret_cast<unsigned long long*>(out + i), (unsigned long long)(v + i));
}

// 64-bit atomic min where one wave instruction covers ROWS rows of 64/ROWS keys (row pitch 2048 keys = 16 KiB): how does the
// rate depend on the shape of the 512 bytes a wave touches?  (k_raster_big's giant path: 8 x 8 px chunks = 8 rows of 64 B)
template <int ROWS>
__global__ __launch_bounds__(256) void cal_atomic8_shape(uint64_t* __restrict__ out, uint64_t v) {
    constexpr int COLS = 64 / ROWS;
    const uint32_t lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwave = (size_t)gridDim.x * 4;
    // the buffer as a 2048-key-wide image of 65536 rows (1 GiB); tiles of ROWS x COLS keys, row-major over the image
    const size_t tiles_x = 2048 / COLS, n_tiles = tiles_x * (65536 / ROWS);
    for (size_t t = wave; t < n_tiles; t += nwave) {
        const size_t ty = t / tiles_x, tx = t - ty * tiles_x;
        const size_t at = (ty * ROWS + lane / COLS) * 2048 + tx * COLS + lane % COLS;
        atomicMin(reinterpret_cast<unsigned long long*>(out + at), (unsigned long long)(v + at));
    }
}
static void run_atomic_shapes() {
    const size_t bytes = (size_t)1 << 30;
    void* b = nullptr;
    CK(hipMalloc(&b, bytes));
    CK(hipMemset(b, 0xFF, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("{\"atomic_min_u64_shapes\": {");
    auto run = [&](const char* name, auto kern, bool last) {
        float best = 1e9f;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(kern, dim3(8192), dim3(256), 0, 0, (uint64_t*)b, 7ull);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        printf("\"%s\": {\"ms_per_GiB\": %.4f, \"G_keys_per_s\": %.2f}%s", name, best, (double)(bytes / 8) / best / 1e6, last ? "" : ", ");
    };
    run("1x64 (512 B run)", cal_atomic8_shape<1>, false);
    run("2x32 (256 B runs)", cal_atomic8_shape<2>, false);
    run("4x16 (128 B runs)", cal_atomic8_shape<4>, false);
    run("8x8 (64 B runs)", cal_atomic8_shape<8>, false);
    run("16x4 (32 B runs)", cal_atomic8_shape<16>, true);
    printf("}}\n");
    CK(hipFree(b));
}

static void run_hbm() {
    const size_t bytes = (size_t)1 << 30;      // 1 GiB: four times the 256 MiB Infinity Cache
    void *a = nullptr, *b = nullptr;
    uint32_t* sink = nullptr;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&b, bytes));
    CK(hipMalloc((void**)&sink, 64));
    CK(hipMemset(a, 0x5A, bytes));
    CK(hipMemset(b, 0xFF, bytes));
    CK(hipDeviceSynchronize());
    const int reps = 3;
    for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(cal_read<uint32_t>, dim3(8192), dim3(256), 0, 0, (const uint32_t*)a, bytes / 4, sink);
        hipLaunchKernelGGL(cal_write<uint32_t>, dim3(8192), dim3(256), 0, 0, (uint32_t*)b, bytes / 4, 0xA5A5A5A5u);      // flushes `a` out of the caches
        hipLaunchKernelGGL(cal_read<uint64_t>, dim3(8192), dim3(256), 0, 0, (const uint64_t*)a, bytes / 8, sink);
        hipLaunchKernelGGL(cal_write<uint64_t>, dim3(8192), dim3(256), 0, 0, (uint64_t*)b, bytes / 8, 0xA5A5A5A5A5A5A5A5ull);
        hipLaunchKernelGGL(cal_read<uint4>, dim3(8192), dim3(256), 0, 0, (const uint4*)a, bytes / 16, sink);
        hipLaunchKernelGGL(cal_write<uint4>, dim3(8192), dim3(256), 0, 0, (uint4*)b, bytes / 16, make_uint4(1, 2, 3, 4));
        // 8 views of 2048 x 4096 keys = 512 MiB, the c4 visibility buffer
        hipLaunchKernelGGL(cal_read8_tile, dim3(2048 / 64, 4096 / 16, 8), dim3(256), 0, 0, (const uint64_t*)a, 2048, 4096, sink);
        hipLaunchKernelGGL(cal_atomic8, dim3(8192), dim3(256), 0, 0, (uint64_t*)b, bytes / 8, 7ull);
    }
    CK(hipDeviceSynchronize());
    printf("{\"hbm_calibration_bytes\": {\"cal_read<unsigned int>\": %zu, \"cal_read<unsigned long>\": %zu, \"cal_read<HIP_vector_type<unsigned int, 4u>>\": %zu, "
           "\"cal_read8_tile\": %zu, \"cal_write<unsigned int>\": %zu, \"cal_write<unsigned long>\": %zu, \"cal_write<HIP_vector_type<unsigned int, 4u>>\": %zu, "
           "\"cal_atomic8\": %zu}}\n",
           bytes, bytes, bytes, (size_t)8 * 2048 * 4096 * 8, bytes, bytes, bytes, bytes);
    CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(sink));
}

// ---- VALU issue cost -----------------------------------------------------------------------------------------------
// Each kernel: kIters iterations of 8 independent instructions of one kind (8 accumulators: no dependency stalls), bracketed
// by s_memtime; the host divides the elapsed shader cycles by the wave-instructions one SIMD issued in that window.
constexpr int kIters = 4096;

#define STREAM8(ASM)                                                                                                   \
    for (int it = 0; it < kIters; ++it) {                                                                              \
        asm volatile(ASM(0) "\n" ASM(1) "\n" ASM(2) "\n" ASM(3) "\n" ASM(4) "\n" ASM(5) "\n" ASM(6) "\n" ASM(7)      \
                     : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])  \
                     : "v"(x), "v"(y));                                                                                \
    }

#define A_FMA(k) "v_fma_f32 %" #k ", %8, %9, %" #k
#define A_MUL(k) "v_mul_f32 %" #k ", %8, %" #k
#define A_ADD(k) "v_add_f32 %" #k ", %8, %" #k
#define A_MULLO(k) "v_mul_lo_u32 %" #k ", %8, %" #k
#define A_MULHI(k) "v_mul_hi_u32 %" #k ", %8, %" #k
#define A_MUL24(k) "v_mul_i32_i24 %" #k ", %8, %" #k
#define A_MAD24(k) "v_mad_i32_i24 %" #k ", %8, %9, %" #k
#define A_ADDU(k) "v_add_u32 %" #k ", %8, %" #k
#define A_RCP(k) "v_rcp_f32 %" #k ", %" #k
#define A_SQRT(k) "v_sqrt_f32 %" #k ", %" #k
#define A_CVT(k) "v_cvt_f32_i32 %" #k ", %" #k
#define A_CVTU(k) "v_cvt_u32_f32 %" #k ", %" #k
#define A_FLOOR(k) "v_floor_f32 %" #k ", %" #k
#define A_FRACT(k) "v_fract_f32 %" #k ", %" #k
#define A_MOV(k) "v_mov_b32 %" #k ", %8"
#define A_MED3(k) "v_med3_f32 %" #k ", %" #k ", %8, %9"
#define A_MAX(k) "v_max_f32 %" #k ", %8, %" #k
#define A_CMP(k) "v_cmp_lt_f32 vcc, %8, %" #k "\nv_cndmask_b32 %" #k ", %8, %9, vcc"
#define A_LSHL(k) "v_lshlrev_b32 %" #k ", 1, %" #k
#define A_AND(k) "v_and_b32 %" #k ", %8, %" #k
#define A_ADD64(k) "v_add_co_u32 %" #k ", vcc, %8, %" #k "\nv_addc_co_u32 %" #k ", vcc, %9, %" #k ", vcc"
#define A_PKFMA(k) "v_pk_fma_f32 %" #k ", %8, %9, %" #k

struct Stamp { uint64_t t0, t1; };

#define VALU_KERNEL(NAME, ASM, TYPE)                                                                                   \
    __global__ __launch_bounds__(256) void NAME(Stamp* __restrict__ st, TYPE x, TYPE y, TYPE* __restrict__ out) {     \
        TYPE r[8];                                                                                                     \
        for (int k = 0; k < 8; ++k) r[k] = x + (TYPE)(threadIdx.x + k);                                                \
        __syncthreads();                                                                                               \
        const uint64_t t0 = __builtin_amdgcn_s_memtime();                                                              \
        STREAM8(ASM)                                                                                                   \
        asm volatile("s_nop 0" ::: "memory");                                                                          \
        const uint64_t t1 = __builtin_amdgcn_s_memtime();                                                              \
        TYPE acc = r[0];                                                                                               \
        for (int k = 1; k < 8; ++k) acc += r[k];                                                                       \
        if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{t0, t1};                          \
        if (acc == (TYPE)12345) out[0] = acc;                                                                          \
    }

VALU_KERNEL(v_fma, A_FMA, float)
VALU_KERNEL(v_mul, A_MUL, float)
VALU_KERNEL(v_add, A_ADD, float)
VALU_KERNEL(v_mullo, A_MULLO, uint32_t)
VALU_KERNEL(v_mulhi, A_MULHI, uint32_t)
VALU_KERNEL(v_mul24, A_MUL24, uint32_t)
VALU_KERNEL(v_mad24, A_MAD24, uint32_t)
VALU_KERNEL(v_addu, A_ADDU, uint32_t)
VALU_KERNEL(v_rcp, A_RCP, float)
VALU_KERNEL(v_sqrt, A_SQRT, float)
VALU_KERNEL(v_cvt_f32_i32, A_CVT, float)
VALU_KERNEL(v_cvt_u32_f32, A_CVTU, float)
VALU_KERNEL(v_floor, A_FLOOR, float)
VALU_KERNEL(v_fract, A_FRACT, float)
VALU_KERNEL(v_mov, A_MOV, float)
VALU_KERNEL(v_med3, A_MED3, float)
VALU_KERNEL(v_max, A_MAX, float)
VALU_KERNEL(v_cmp_cndmask, A_CMP, float)
VALU_KERNEL(v_lshl, A_LSHL, uint32_t)
VALU_KERNEL(v_and, A_AND, uint32_t)
VALU_KERNEL(v_add64, A_ADD64, uint32_t)

// packed f32 (two f32 operations per lane and instruction, operands in aligned register pairs)
typedef float v2f __attribute__((ext_vector_type(2)));
#define A_PKMUL(k) "v_pk_mul_f32 %" #k ", %8, %" #k
#define A_PKADD(k) "v_pk_add_f32 %" #k ", %8, %" #k
#define PK_KERNEL(NAME, ASM)                                                                                           \
    __global__ __launch_bounds__(256) void NAME(Stamp* __restrict__ st, float xs, float ys, float* __restrict__ out) { \
        v2f r[8];                                                                                                      \
        const v2f x = {xs, xs * 1.5f}, y = {ys, ys * 0.5f};                                                            \
        for (int k = 0; k < 8; ++k) r[k] = v2f{xs + (float)(threadIdx.x + k), xs - (float)k};                           \
        __syncthreads();                                                                                               \
        const uint64_t t0 = __builtin_amdgcn_s_memtime();                                                              \
        STREAM8(ASM)                                                                                                   \
        asm volatile("s_nop 0" ::: "memory");                                                                          \
        const uint64_t t1 = __builtin_amdgcn_s_memtime();                                                              \
        float acc = 0.0f;                                                                                              \
        for (int k = 0; k < 8; ++k) acc += r[k].x + r[k].y;                                                            \
        if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{t0, t1};                          \
        if (acc == 12345.0f) out[0] = acc;                                                                             \
    }
PK_KERNEL(v_pkfma, A_PKFMA)
PK_KERNEL(v_pkmul, A_PKMUL)
PK_KERNEL(v_pkadd, A_PKADD)

// 64-bit integer forms (register pairs): what the int64 edge functions of k_raster_big's giant path compile to
typedef uint64_t u64;
#define A_LSHLADD64(k) "v_lshl_add_u64 %" #k ", %" #k ", 0, %8"
#define A_MAD64(k) "v_mad_u64_u32 %" #k ", vcc, %10, %11, %" #k
#define A_CMP64(k) "v_cmp_lt_i64 vcc, %8, %" #k
#define A_SHL64(k) "v_lshlrev_b64 %" #k ", 1, %" #k
#define I64_KERNEL(NAME, ASM)                                                                                          \
    __global__ __launch_bounds__(256) void NAME(Stamp* __restrict__ st, uint32_t xs, uint32_t ys, uint32_t* __restrict__ out) { \
        u64 r[8];                                                                                                      \
        const u64 x = ((u64)xs << 32) | ys, y = ((u64)ys << 20) | xs;                                                  \
        const uint32_t a = xs + threadIdx.x, b = ys | 1u;                                                              \
        for (int k = 0; k < 8; ++k) r[k] = x + (u64)(threadIdx.x + k) * 0x100000001ull;                                \
        __syncthreads();                                                                                               \
        const uint64_t t0 = __builtin_amdgcn_s_memtime();                                                              \
        for (int it = 0; it < kIters; ++it) {                                                                          \
            asm volatile(ASM(0) "\n" ASM(1) "\n" ASM(2) "\n" ASM(3) "\n" ASM(4) "\n" ASM(5) "\n" ASM(6) "\n" ASM(7)      \
                         : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])  \
                         : "v"(x), "v"(y), "v"(a), "v"(b) : "vcc");                                                    \
        }                                                                                                              \
        asm volatile("s_nop 0" ::: "memory");                                                                          \
        const uint64_t t1 = __builtin_amdgcn_s_memtime();                                                              \
        u64 acc = 0;                                                                                                   \
        for (int k = 0; k < 8; ++k) acc += r[k];                                                                       \
        if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{t0, t1};                          \
        if (acc == 12345ull) out[0] = (uint32_t)acc;                                                                   \
    }
// scalar unit: the same stream shape on SGPRs (one scalar ALU serves the CU's four SIMDs)
#define S_ADD(k) "s_add_u32 %" #k ", %" #k ", %8"
#define S_MUL(k) "s_mul_i32 %" #k ", %" #k ", %8"
#define S_MULHI(k) "s_mul_hi_u32 %" #k ", %" #k ", %8"
#define S_AND64(k) "s_and_b64 vcc, exec, vcc\ns_add_u32 %" #k ", %" #k ", %8"
#define S_CSEL(k) "s_cmp_lt_u32 %" #k ", %9\ns_cselect_b32 %" #k ", %8, %" #k
#define S_BR(k) "s_cmp_lt_u32 %" #k ", %9\ns_cbranch_scc1 1f\ns_add_u32 %" #k ", %" #k ", %8\n1:"
#define SALU_KERNEL(NAME, ASM)                                                                                         \
    __global__ __launch_bounds__(256) void NAME(Stamp* __restrict__ st, uint32_t xs, uint32_t ys, uint32_t* __restrict__ out) { \
        uint32_t r[8];                                                                                                 \
        for (int k = 0; k < 8; ++k) r[k] = __builtin_amdgcn_readfirstlane(xs + (threadIdx.x >> 6) + k);                \
        const uint32_t x = __builtin_amdgcn_readfirstlane(xs | 1u), y = __builtin_amdgcn_readfirstlane(ys);            \
        __syncthreads();                                                                                               \
        const uint64_t t0 = __builtin_amdgcn_s_memtime();                                                              \
        for (int it = 0; it < kIters; ++it) {                                                                          \
            asm volatile(ASM(0) "\n" ASM(1) "\n" ASM(2) "\n" ASM(3) "\n" ASM(4) "\n" ASM(5) "\n" ASM(6) "\n" ASM(7)      \
                         : "+s"(r[0]), "+s"(r[1]), "+s"(r[2]), "+s"(r[3]), "+s"(r[4]), "+s"(r[5]), "+s"(r[6]), "+s"(r[7])  \
                         : "s"(x), "s"(y) : "vcc", "scc");                                                             \
        }                                                                                                              \
        asm volatile("s_nop 0" ::: "memory");                                                                          \
        const uint64_t t1 = __builtin_amdgcn_s_memtime();                                                              \
        uint32_t acc = 0;                                                                                              \
        for (int k = 0; k < 8; ++k) acc += r[k];                                                                       \
        if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{t0, t1};                          \
        if (acc == 12345u) out[0] = acc;                                                                               \
    }
SALU_KERNEL(s_add, S_ADD)
SALU_KERNEL(s_mul, S_MUL)
SALU_KERNEL(s_mulhi, S_MULHI)
SALU_KERNEL(s_and64_add, S_AND64)
SALU_KERNEL(s_cmp_csel, S_CSEL)
SALU_KERNEL(s_cmp_br_add, S_BR)
// do vector and scalar instructions share the SIMD's issue?  8 vector + 8 scalar instructions per iteration, interleaved
#define MIX_KERNEL(NAME, VOP, SOP)                                                                                     \
    __global__ __launch_bounds__(256) void NAME(Stamp* __restrict__ st, float xf, float yf, float* __restrict__ out) { \
        float r[8];                                                                                                    \
        uint32_t q[8];                                                                                                 \
        for (int k = 0; k < 8; ++k) { r[k] = xf + (float)(threadIdx.x + k); q[k] = __builtin_amdgcn_readfirstlane((uint32_t)k + (threadIdx.x >> 6)); } \
        const uint32_t sx = __builtin_amdgcn_readfirstlane(__float_as_uint(xf) | 1u);                                  \
        __syncthreads();                                                                                               \
        const uint64_t t0 = __builtin_amdgcn_s_memtime();                                                              \
        for (int it = 0; it < kIters; ++it) {                                                                          \
            asm volatile(VOP(0) "\n" SOP(8) "\n" VOP(1) "\n" SOP(9) "\n" VOP(2) "\n" SOP(10) "\n" VOP(3) "\n" SOP(11) "\n"    \
                         VOP(4) "\n" SOP(12) "\n" VOP(5) "\n" SOP(13) "\n" VOP(6) "\n" SOP(14) "\n" VOP(7) "\n" SOP(15)        \
                         : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),  \
                           "+s"(q[0]), "+s"(q[1]), "+s"(q[2]), "+s"(q[3]), "+s"(q[4]), "+s"(q[5]), "+s"(q[6]), "+s"(q[7])   \
                         : "v"(xf), "v"(yf), "s"(sx) : "scc");                                                         \
        }                                                                                                              \
        asm volatile("s_nop 0" ::: "memory");                                                                          \
        const uint64_t t1 = __builtin_amdgcn_s_memtime();                                                              \
        float acc = 0.0f;                                                                                              \
        for (int k = 0; k < 8; ++k) acc += r[k] + (float)q[k];                                                         \
        if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{t0, t1};                          \
        if (acc == 12345.0f) out[0] = acc;                                                                             \
    }
#define MV_FMA(k) "v_fma_f32 %" #k ", %16, %17, %" #k
#define MV_MAX(k) "v_max_f32 %" #k ", %16, %" #k
#define MS_ADD(k) "s_add_u32 %" #k ", %" #k ", %18"
#define MS_MUL(k) "s_mul_i32 %" #k ", %" #k ", %18"
MIX_KERNEL(mix_valu_salu, MV_FMA, MS_ADD)
MIX_KERNEL(mix_half_salu, MV_MAX, MS_ADD)
MIX_KERNEL(mix_fma_smul, MV_FMA, MS_MUL)
I64_KERNEL(v_lshladd64, A_LSHLADD64)
I64_KERNEL(v_mad64, A_MAD64)
I64_KERNEL(v_cmp64, A_CMP64)
I64_KERNEL(v_shl64, A_SHL64)

template <typename K, typename T>
static void time_valu(const char* name, K kernel, T x, T y, int insts_per_slot, std::string& json) {
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    Stamp* st = nullptr;
    T* out = nullptr;
    CK(hipMalloc((void**)&out, 64));
    double res[2] = {0, 0}, ns[2] = {0, 0}, span[2] = {0, 0}, wall[2] = {0, 0};
    const int occ[2] = {1, 8};
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int o = 0; o < 2; ++o) {
        const int wgs = cus * occ[o];           // 4 waves per workgroup = one per SIMD; occ[o] workgroups per CU
        CK(hipMalloc((void**)&st, sizeof(Stamp) * wgs * 4));
        hipLaunchKernelGGL(kernel, dim3(wgs), dim3(256), 0, 0, st, x, y, out);
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kernel, dim3(wgs), dim3(256), 0, 0, st, x, y, out);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0.0f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ns[o] = (double)ms * 1e6 / ((double)kIters * 8 * insts_per_slot * occ[o]);      // wall time per wave-instruction of one SIMD
        std::vector<Stamp> h(wgs * 4);
        CK(hipMemcpy(h.data(), st, sizeof(Stamp) * wgs * 4, hipMemcpyDeviceToHost));
        // median elapsed cycles of a wave; under full occupancy all occ[o] waves of a SIMD run for that whole window
        std::vector<double> el;
        for (auto& s : h) el.push_back((double)(s.t1 - s.t0));
        std::sort(el.begin(), el.end());
        const double med = el[el.size() / 2];
        uint64_t tmin = ~0ull, tmax = 0;
        for (auto& s : h) { tmin = s.t0 < tmin ? s.t0 : tmin; tmax = s.t1 > tmax ? s.t1 : tmax; }
        span[o] = (double)(tmax - tmin);       // ticks from the first wave's start to the last wave's end
        wall[o] = (double)ms * 1e6;
        res[o] = med / ((double)kIters * 8 * insts_per_slot * occ[o]);
        CK(hipFree(st));
    }
    char buf[512];
    snprintf(buf, sizeof buf, "%s\"%s\": {\"cyc_per_inst_1wave\": %.3f, \"cyc_per_inst_8waves\": %.3f, \"ns_per_inst_1wave\": %.4f, \"ns_per_inst_8waves\": %.4f, \"ticks_per_ns_8waves\": %.4f, \"wave_window_over_kernel_span_8waves\": %.3f}",
             json.empty() ? "" : ", ", name, res[0], res[1], ns[0], ns[1], span[1] / wall[1], res[1] * kIters * 8 * insts_per_slot * 8 / span[1]);
    json += buf;
    CK(hipFree(out));
}

#include <algorithm>

static void run_valu() {
    std::string j;
    time_valu("v_fma_f32", v_fma, 1.0001f, 0.5f, 1, j);
    time_valu("v_mul_f32", v_mul, 1.0001f, 0.5f, 1, j);
    time_valu("v_add_f32", v_add, 1.0001f, 0.5f, 1, j);
    time_valu("v_mul_lo_u32", v_mullo, 3u, 5u, 1, j);
    time_valu("v_mul_hi_u32", v_mulhi, 3u, 5u, 1, j);
    time_valu("v_mul_i32_i24", v_mul24, 3u, 5u, 1, j);
    time_valu("v_mad_i32_i24", v_mad24, 3u, 5u, 1, j);
    time_valu("v_add_u32", v_addu, 3u, 5u, 1, j);
    time_valu("v_rcp_f32", v_rcp, 1.5f, 0.5f, 1, j);
    time_valu("v_sqrt_f32", v_sqrt, 1.5f, 0.5f, 1, j);
    time_valu("v_cvt_f32_i32", v_cvt_f32_i32, 1.5f, 0.5f, 1, j);
    time_valu("v_cvt_u32_f32", v_cvt_u32_f32, 1.5f, 0.5f, 1, j);
    time_valu("v_floor_f32", v_floor, 1.5f, 0.5f, 1, j);
    time_valu("v_fract_f32", v_fract, 1.5f, 0.5f, 1, j);
    time_valu("v_mov_b32", v_mov, 1.5f, 0.5f, 1, j);
    time_valu("v_med3_f32", v_med3, 1.5f, 0.5f, 1, j);
    time_valu("v_max_f32", v_max, 1.5f, 0.5f, 1, j);
    time_valu("v_cmp+v_cndmask", v_cmp_cndmask, 1.5f, 0.5f, 2, j);
    time_valu("v_lshlrev_b32", v_lshl, 3u, 5u, 1, j);
    time_valu("v_and_b32", v_and, 3u, 5u, 1, j);
    time_valu("v_add_co+v_addc (64-bit add)", v_add64, 3u, 5u, 2, j);
    time_valu("v_pk_fma_f32", v_pkfma, 1.0001f, 0.5f, 1, j);
    time_valu("v_pk_mul_f32", v_pkmul, 1.0001f, 0.5f, 1, j);
    time_valu("v_pk_add_f32", v_pkadd, 1.0001f, 0.5f, 1, j);
    time_valu("s_add_u32", s_add, 3u, 5u, 1, j);
    time_valu("s_mul_i32", s_mul, 3u, 5u, 1, j);
    time_valu("s_mul_hi_u32", s_mulhi, 3u, 5u, 1, j);
    time_valu("s_and_b64+s_add_u32", s_and64_add, 3u, 5u, 2, j);
    time_valu("s_cmp+s_cselect", s_cmp_csel, 3u, 5u, 2, j);
    time_valu("s_cmp+s_cbranch(not taken)+s_add", s_cmp_br_add, 3u, 0u, 3, j);
    time_valu("s_cmp+s_cbranch(taken)", s_cmp_br_add, 3u, 0xFFFFFFFFu, 2, j);
    time_valu("mix: v_fma_f32+s_add_u32 (per pair)", mix_valu_salu, 1.0001f, 0.5f, 1, j);
    time_valu("mix: v_max_f32+s_add_u32 (per pair)", mix_half_salu, 1.0001f, 0.5f, 1, j);
    time_valu("mix: v_fma_f32+s_mul_i32 (per pair)", mix_fma_smul, 1.0001f, 0.5f, 1, j);
    time_valu("v_lshl_add_u64", v_lshladd64, 3u, 5u, 1, j);
    time_valu("v_mad_u64_u32", v_mad64, 3u, 5u, 1, j);
    time_valu("v_cmp_lt_i64", v_cmp64, 3u, 5u, 1, j);
    time_valu("v_lshlrev_b64", v_shl64, 3u, 5u, 1, j);
    int clk = 0;
    CK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0));
    printf("{\"valu_issue_cost\": {%s}, \"unit\": \"cyc_*: s_memtime ticks, ns_*: HIP-event wall time (incl. ~10 us of launch), per wave64 instruction per SIMD\", \"clock_rate_khz\": %d}\n", j.c_str(), clk);
}

// ---- copy: what a pure read + write stream of the load phase's size reaches on this device (the ceiling k_normals_* is held to)
typedef float cal_f4 __attribute__((ext_vector_type(4)));
template <bool kNt>
__global__ __launch_bounds__(256) void cal_copy16(const cal_f4* __restrict__ in, cal_f4* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const cal_f4 v = in[i];
        if (kNt) __builtin_nontemporal_store(v, out + i);
        else out[i] = v;
    }
}
static void run_copy() {
    const size_t bytes = (size_t)576 << 20;      // the c4 DEM: 100 tiles of 1200 x 1200 f32 (and as many bytes of normals out)
    void *a = nullptr, *b = nullptr;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("{\"copy_576MiB\": {");
    auto run = [&](const char* name, auto kern, unsigned grid, bool last) {
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, (const cal_f4*)a, (cal_f4*)b, bytes / 16);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        printf("\"%s\": {\"ms\": %.4f, \"read_plus_write_TBps\": %.3f}%s", name, best, 2.0 * bytes / best / 1e9, last ? "" : ", ");
    };
    run("plain stores, grid 2048", cal_copy16<false>, 2048, false);
    run("plain stores, grid 8192", cal_copy16<false>, 8192, false);
    run("plain stores, one 16 B per thread", cal_copy16<false>, (unsigned)(bytes / 16 / 256), false);
    run("non-temporal stores, grid 2048", cal_copy16<true>, 2048, false);
    run("non-temporal stores, grid 8192", cal_copy16<true>, 8192, false);
    run("non-temporal stores, one 16 B per thread", cal_copy16<true>, (unsigned)(bytes / 16 / 256), true);
    printf("}}\n");
    CK(hipFree(a)); CK(hipFree(b));
}

int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "copy")) { run_copy(); return 0; }
    if (argc > 1 && !strcmp(argv[1], "hbm")) run_hbm();
    else if (argc > 1 && !strcmp(argv[1], "valu")) run_valu();
    else if (argc > 1 && !strcmp(argv[1], "atomics")) run_atomic_shapes();
    else { fprintf(stderr, "usage: calib hbm|valu|atomics\n"); return 2; }
    return 0;
}
