// valu_rate_probe.hip -- measurement aid (VERDICT round 2 "weak" #1): the ISSUE RATE of the wave64
// instructions the instruction-bound kernels of this repo are made of, measured on the MI355X it runs on.
//
// The question it settles: how many cycles does one SIMD need per wave64 VALU instruction?  The microarchitecture guide's
// wave-scheduling section says 2 (SIMD-32); bench.py's `valu_issue_frac` assumes 4 (16 lanes per cycle for non-packed
// integer operations).  Every kernel below is a loop of 64 instructions of ONE kind on 8 independent registers
// (a dependent chain of length 8 apart), run by w = 1, 2, 4, 8 waves per SIMD on every SIMD of the chip (one 256-thread
// workgroup = one wave per SIMD; the dynamic LDS size lets exactly w workgroups share a CU).  Two clocks:
//   * per wave:  s_memtime around the loop  -> shader cycles per instruction as one wave sees it,
//   * per chip:  hipEvents around the launch -> wave-instructions per second = (cycles per instruction per SIMD)^-1 at
//                the clock the chip held during the launch (printed: s_memtime ticks / event time).
// cycles per instruction per SIMD = (per-wave cycles per instruction) / w once the pipe is full.
//
// Build + run:  hipcc --offload-arch=gfx950 -O2 -o valu_rate_probe valu_rate_probe.hip && ./valu_rate_probe [json path]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

constexpr int ITERS = 2048;       // loop trips; 64 instructions each

// one instruction per register r0..r7 (%0..%7), eight times over, in ONE asm statement (between two asm statements the
// compiler pads with s_nop, which would be measured too); %8 / %9 = loop-invariant operands
#define B8(I) I("%0") I("%1") I("%2") I("%3") I("%4") I("%5") I("%6") I("%7")
#define B64(I) B8(I) B8(I) B8(I) B8(I) B8(I) B8(I) B8(I) B8(I)
#define I_v_add_u32(R) "v_add_u32 " R ", " R ", %8" "\n"
#define I_v_and_b32(R) "v_and_b32 " R ", " R ", %8" "\n"
#define I_v_lshl_add_u32(R) "v_lshl_add_u32 " R ", " R ", 1, %8" "\n"
#define I_v_add3_u32(R) "v_add3_u32 " R ", " R ", %8, %9" "\n"
#define I_v_max3_u32(R) "v_max3_u32 " R ", " R ", %8, %9" "\n"
#define I_v_bfe_u32(R) "v_bfe_u32 " R ", " R ", 3, 9" "\n"
#define I_v_perm_b32(R) "v_perm_b32 " R ", " R ", %8, %9" "\n"
#define I_v_alignbyte_b32(R) "v_alignbyte_b32 " R ", " R ", %8, 3" "\n"
#define I_v_cndmask_b32(R) "v_cndmask_b32 " R ", " R ", %8, vcc" "\n"
#define I_v_cmp_lt_u32(R) "v_cmp_lt_u32 vcc, " R ", %8" "\n"
#define I_v_cmp_ne_u16_sdwa(R) "v_cmp_ne_u16_sdwa vcc, " R ", %8 src0_sel:DWORD src1_sel:BYTE_1" "\n"
#define I_v_cmpx_ne_u16_sdwa(R) "v_cmpx_ne_u16_sdwa vcc, %9, %8 src0_sel:DWORD src1_sel:BYTE_1" "\n"
#define I_v_add_u32_dpp_row_shr(R) "v_add_u32_dpp " R ", " R ", %8 row_shr:1 row_mask:0xf bank_mask:0xf" "\n"
#define I_v_sub_co_u32_dpp(R) "v_sub_co_u32_dpp " R ", vcc, " R ", %8 row_shr:1 row_mask:0xf bank_mask:0xf" "\n"
#define I_v_addc_co_u32(R) "v_addc_co_u32 " R ", vcc, " R ", %8, vcc" "\n"
#define I_v_mul_lo_u32(R) "v_mul_lo_u32 " R ", " R ", %8" "\n"
#define I_v_mul_hi_u32(R) "v_mul_hi_u32 " R ", " R ", %8" "\n"
#define I_v_mul_u32_u24(R) "v_mul_u32_u24 " R ", " R ", %8" "\n"
#define I_v_mad_u32_u24(R) "v_mad_u32_u24 " R ", " R ", %8, %9" "\n"
#define I_v_mad_u64_u32(R) "v_mad_u64_u32 v[40:41], vcc, " R ", %8, v[40:41]" "\n"
#define I_v_lshlrev_b64(R) "v_lshlrev_b64 v[40:41], 1, v[40:41]" "\n"
#define I_v_bcnt_u32_b32(R) "v_bcnt_u32_b32 " R ", " R ", %8" "\n"
#define I_v_mbcnt_lo(R) "v_mbcnt_lo_u32_b32 " R ", %8, " R "" "\n"
#define I_v_readlane(R) "v_readlane_b32 s20, " R ", 3" "\n"
#define I_v_pk_add_u16(R) "v_pk_add_u16 " R ", " R ", %8" "\n"
#define I_v_mov_b32(R) "v_mov_b32 " R ", %8" "\n"
#define I_s_add_u32(R) "s_add_u32 s20, s20, s21" "\n"
#define I_s_and_b64(R) "s_and_b64 s[20:21], s[20:21], s[22:23]" "\n"
#define I_s_bfe_u32_indep(R) "s_bfe_u32 s20, s21, 0x50003" "\n"
#define I_v_or_b32(R) "v_or_b32 " R ", " R ", %8" "\n"
#define I_v_xor_b32(R) "v_xor_b32 " R ", " R ", %8" "\n"
#define I_v_sub_u32(R) "v_sub_u32 " R ", " R ", %8" "\n"
#define I_v_lshlrev_b32(R) "v_lshlrev_b32 " R ", 1, " R "" "\n"
#define I_v_lshrrev_b32(R) "v_lshrrev_b32 " R ", 1, " R "" "\n"
#define I_v_min_u32(R) "v_min_u32 " R ", " R ", %8" "\n"
#define I_v_max_u32(R) "v_max_u32 " R ", " R ", %8" "\n"
#define I_v_add_co_u32(R) "v_add_co_u32 " R ", vcc, " R ", %8" "\n"
#define I_v_add_u32_e64(R) "v_add_u32_e64 " R ", " R ", %8" "\n"
#define I_v_add_u32_sgpr(R) "v_add_u32 " R ", s21, " R "" "\n"
#define I_v_add_u32_lit(R) "v_add_u32 " R ", 0x12345, " R "" "\n"
#define I_v_add_u32_sdwa(R) "v_add_u32_sdwa " R ", " R ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" "\n"
#define I_v_and_or_b32(R) "v_and_or_b32 " R ", " R ", %8, %9" "\n"
#define I_v_or3_b32(R) "v_or3_b32 " R ", " R ", %8, %9" "\n"
#define I_v_add_lshl_u32(R) "v_add_lshl_u32 " R ", " R ", %8, 1" "\n"
#define I_v_xad_u32(R) "v_xad_u32 " R ", " R ", %8, %9" "\n"
#define I_v_cndmask_b32_e64_sgpr(R) "v_cndmask_b32_e64 " R ", " R ", %8, s[22:23]" "\n"
#define I_v_cndmask_b32_e64_vcc(R) "v_cndmask_b32_e64 " R ", " R ", %8, vcc" "\n"
#define I_v_cndmask_b32_other_dst(R) "v_cndmask_b32 v40, " R ", %8, vcc" "\n"
#define I_v_cmp_then_cndmask(R) "v_cmp_lt_u32 vcc, " R ", %8\nv_cndmask_b32 " R ", " R ", %9, vcc" "\n"
#define I_v_cmp_e64_then_cndmask_e64(R) "v_cmp_lt_u32_e64 s[22:23], " R ", %8\nv_cndmask_b32_e64 " R ", " R ", %9, s[22:23]" "\n"
#define I_v_and_b32_dep_chain(R) "v_and_b32 %0, %0, %8" "\n"
#define I_v_mov_b32_dpp_row_shr(R) "v_mov_b32_dpp " R ", " R " row_shr:1 row_mask:0xf bank_mask:0xf" "\n"
#define I_v_add_u32_dpp_row_bcast(R) "v_add_u32_dpp " R ", " R ", %8 row_bcast:15 row_mask:0xa bank_mask:0xf" "\n"
#define I_v_bfi_b32(R) "v_bfi_b32 " R ", " R ", %8, %9" "\n"
#define I_v_lshl_or_b32(R) "v_lshl_or_b32 " R ", " R ", 3, %8" "\n"
#define I_v_cvt_f32_u32(R) "v_cvt_f32_u32 " R ", " R "" "\n"
#define I_v_fma_f32(R) "v_fma_f32 " R ", " R ", %8, %9" "\n"
#define I_v_add_f32(R) "v_add_f32 " R ", " R ", %8" "\n"
#define I_v_pk_fma_f32(R) "v_pk_fma_f32 v[40:41], v[40:41], v[40:41], v[40:41]" "\n"
#define I_s_nop_0(R) "s_nop 0" "\n"
#define I_s_mov_b32(R) "s_mov_b32 s20, s21" "\n"
#define OP_LIST(X) X(v_add_u32) X(v_and_b32) X(v_lshl_add_u32) X(v_add3_u32) X(v_max3_u32) X(v_bfe_u32) X(v_perm_b32) X(v_alignbyte_b32) X(v_cndmask_b32) X(v_cmp_lt_u32) X(v_cmp_ne_u16_sdwa) X(v_cmpx_ne_u16_sdwa) X(v_add_u32_dpp_row_shr) X(v_sub_co_u32_dpp) X(v_addc_co_u32) X(v_mul_lo_u32) X(v_mul_hi_u32) X(v_mul_u32_u24) X(v_mad_u32_u24) X(v_mad_u64_u32) X(v_lshlrev_b64) X(v_bcnt_u32_b32) X(v_mbcnt_lo) X(v_readlane) X(v_pk_add_u16) X(v_mov_b32) X(s_add_u32) X(s_and_b64) X(s_bfe_u32_indep) X(v_or_b32) X(v_xor_b32) X(v_sub_u32) X(v_lshlrev_b32) X(v_lshrrev_b32) X(v_min_u32) X(v_max_u32) X(v_add_co_u32) X(v_add_u32_e64) X(v_add_u32_sgpr) X(v_add_u32_lit) X(v_add_u32_sdwa) X(v_and_or_b32) X(v_or3_b32) X(v_add_lshl_u32) X(v_xad_u32) X(v_cndmask_b32_e64_sgpr) X(v_cndmask_b32_e64_vcc) X(v_cndmask_b32_other_dst) X(v_cmp_then_cndmask) X(v_cmp_e64_then_cndmask_e64) X(v_and_b32_dep_chain) X(v_mov_b32_dpp_row_shr) X(v_add_u32_dpp_row_bcast) X(v_bfi_b32) X(v_lshl_or_b32) X(v_cvt_f32_u32) X(v_fma_f32) X(v_add_f32) X(v_pk_fma_f32) X(s_nop_0) X(s_mov_b32)

#define DEF_BODY(name) \
    __global__ __launch_bounds__(256) void k_##name(uint32_t *out, uint64_t *ticks, uint32_t a, uint32_t b) { \
        extern __shared__ uint32_t s_dyn[]; \
        uint32_t r[8]; \
        _Pragma("unroll") for (int i = 0; i < 8; i++) r[i] = threadIdx.x * 8 + i; \
        if (a == 0xFFFFFFFFu) s_dyn[threadIdx.x] = 1; \
        const uint64_t t0 = __builtin_readcyclecounter(); \
        for (int it = 0; it < ITERS; it++) { \
            asm volatile(B64(I_##name) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) \
                         : "v"(a), "v"(b) : "vcc", "scc", "s20", "s21", "s22", "s23", "v40", "v41"); \
        } \
        const uint64_t t1 = __builtin_readcyclecounter(); \
        uint32_t acc = 0; \
        _Pragma("unroll") for (int i = 0; i < 8; i++) acc ^= r[i]; \
        if (acc == 0x12345u) out[0] = acc; \
        if ((threadIdx.x & 63) == 0) ticks[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0; \
    }
OP_LIST(DEF_BODY)

// LDS operations: address = a per-lane conflict-free slot; waited for in groups of 8
#define LDS_LIST(X) \
    X(ds_read_u8, "ds_read_u8 %0, %8", "ds_read_u8 %1, %8", "ds_read_u8 %2, %8", "ds_read_u8 %3, %8", "ds_read_u8 %4, %8", "ds_read_u8 %5, %8", "ds_read_u8 %6, %8", "ds_read_u8 %7, %8", 1) \
    X(ds_read_b32, "ds_read_b32 %0, %8", "ds_read_b32 %1, %8", "ds_read_b32 %2, %8", "ds_read_b32 %3, %8", "ds_read_b32 %4, %8", "ds_read_b32 %5, %8", "ds_read_b32 %6, %8", "ds_read_b32 %7, %8", 4) \
    X(ds_read_b64, "ds_read_b64 %0, %8", "ds_read_b64 %1, %8", "ds_read_b64 %2, %8", "ds_read_b64 %3, %8", "ds_read_b64 %4, %8", "ds_read_b64 %5, %8", "ds_read_b64 %6, %8", "ds_read_b64 %7, %8", 8) \
    X(ds_read_b128, "ds_read_b128 %0, %8", "ds_read_b128 %1, %8", "ds_read_b128 %2, %8", "ds_read_b128 %3, %8", "ds_read_b128 %4, %8", "ds_read_b128 %5, %8", "ds_read_b128 %6, %8", "ds_read_b128 %7, %8", 16) \
    X(ds_write_b32, "ds_write_b32 %8, %9", "ds_write_b32 %8, %9", "ds_write_b32 %8, %9", "ds_write_b32 %8, %9", "ds_write_b32 %8, %9", "ds_write_b32 %8, %9", "ds_write_b32 %8, %9", "ds_write_b32 %8, %9", 4) \
    X(ds_add_rtn_u32, "ds_add_rtn_u32 %0, %8, %9", "ds_add_rtn_u32 %1, %8, %9", "ds_add_rtn_u32 %2, %8, %9", "ds_add_rtn_u32 %3, %8, %9", "ds_add_rtn_u32 %4, %8, %9", "ds_add_rtn_u32 %5, %8, %9", "ds_add_rtn_u32 %6, %8, %9", "ds_add_rtn_u32 %7, %8, %9", 4) \
    X(ds_bpermute_b32, "ds_bpermute_b32 %0, %8, %9", "ds_bpermute_b32 %1, %8, %9", "ds_bpermute_b32 %2, %8, %9", "ds_bpermute_b32 %3, %8, %9", "ds_bpermute_b32 %4, %8, %9", "ds_bpermute_b32 %5, %8, %9", "ds_bpermute_b32 %6, %8, %9", "ds_bpermute_b32 %7, %8, %9", 4)

template <int W> struct VecT;
template <> struct VecT<1> { typedef uint32_t type; };
template <> struct VecT<2> { typedef uint32_t type __attribute__((ext_vector_type(2))); };
template <> struct VecT<4> { typedef uint32_t type __attribute__((ext_vector_type(4))); };
template <int W> using Vec = typename VecT<W>::type;
#define DEF_LDS(name, i0, i1, i2, i3, i4, i5, i6, i7, bytes) \
    __global__ __launch_bounds__(256) void k_##name(uint32_t *out, uint64_t *ticks, uint32_t a, uint32_t b) { \
        extern __shared__ uint32_t s_dyn[]; \
        for (uint32_t i = threadIdx.x; i < 256 * 4; i += 256) s_dyn[i] = i; \
        __syncthreads(); \
        const uint32_t addr = (uint32_t)(threadIdx.x * (bytes < 4 ? 4 : bytes)); \
        const uint64_t t0 = __builtin_readcyclecounter(); \
        for (int it = 0; it < ITERS * 8; it++) { \
            Vec<(bytes + 3) / 4> q[8]; \
            asm volatile(i0 "\n" i1 "\n" i2 "\n" i3 "\n" i4 "\n" i5 "\n" i6 "\n" i7 "\ns_waitcnt lgkmcnt(0)" \
                         : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4]), "=&v"(q[5]), "=&v"(q[6]), "=&v"(q[7]) \
                         : "v"(addr), "v"(b) : "memory"); \
        } \
        const uint64_t t1 = __builtin_readcyclecounter(); \
        if (a == 77) out[0] = 1; \
        if ((threadIdx.x & 63) == 0) ticks[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0; \
    }
LDS_LIST(DEF_LDS)

typedef void (*kern_t)(uint32_t *, uint64_t *, uint32_t, uint32_t);
struct Entry { const char *name; kern_t fn; const char *unit; };

int main(int argc, char **argv)
{
    std::vector<Entry> entries;
#define PUSH_OP(name) entries.push_back({#name, k_##name, "valu/salu"});
    OP_LIST(PUSH_OP)
#define PUSH_LDS(name, i0, i1, i2, i3, i4, i5, i6, i7, bytes) entries.push_back({#name, k_##name, "lds"});
    LDS_LIST(PUSH_LDS)

    setvbuf(stdout, NULL, _IONBF, 0);
    const char *filter = argc > 2 ? argv[2] : nullptr;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int simds = cus * 4;
    uint32_t *d_out; uint64_t *d_ticks;
    CHECK(hipMalloc(&d_out, 64));
    CHECK(hipMalloc(&d_ticks, sizeof(uint64_t) * cus * 8 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::string json = "{\"device\": \"" + std::string(prop.gcnArchName) + "\", \"cus\": " + std::to_string(cus) +
                       ", \"clock_rate_khz\": " + std::to_string(prop.clockRate) +
                       ", \"instructions_per_wave\": " + std::to_string((long long)ITERS * 64) + ", \"ops\": {";
    printf("%-24s %5s %14s %14s %12s %10s\n", "instruction", "w/SIMD", "cyc/inst/wave", "cyc/inst/SIMD", "Ginst/s chip", "tick GHz");
    bool first = true;
    for (const Entry &en : entries) {
        if (filter && !strstr(en.name, filter)) continue;
        json += std::string(first ? "" : ", ") + "\"" + en.name + "\": {";
        first = false;
        const int ws[4] = {1, 2, 4, 8};
        for (int wi = 0; wi < 4; wi++) {
            const int w = ws[wi];
            // dynamic LDS so that exactly w workgroups fit a CU (160 KB per CU; 64 KB cap per workgroup -> w = 1 uses 64 KB
            // and relies on the grid size alone)
            size_t lds = (size_t)(160 * 1024) / (w + 1) + 1024;
            if (lds > 64 * 1024) lds = 64 * 1024;
            if (lds < 8192) lds = 8192;
            CHECK(hipFuncSetAttribute((const void *)en.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const int grid = cus * w;
            hipLaunchKernelGGL(en.fn, dim3(grid), dim3(256), lds, 0, d_out, d_ticks, 1u, 0x01000100u);      // warm-up
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(en.fn, dim3(grid), dim3(256), lds, 0, d_out, d_ticks, 1u, 0x01000100u);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<uint64_t> t(grid * 4);
            CHECK(hipMemcpy(t.data(), d_ticks, sizeof(uint64_t) * grid * 4, hipMemcpyDeviceToHost));
            double mean = 0, mx = 0;
            for (uint64_t v : t) { mean += (double)v; if ((double)v > mx) mx = (double)v; }
            mean /= (double)t.size();
            const double ninst = (double)ITERS * 64;
            const double cyc_wave = mean / ninst, cyc_simd = cyc_wave / w;
            const double ginst = ninst * grid * 4 / (ms * 1e-3) / 1e9;       // wave-instructions per second, whole chip
            const double tick_ghz = mx / (ms * 1e-3) / 1e9;                   // longest wave ~ the launch
            printf("%-24s %5d %14.3f %14.3f %12.1f %10.3f\n", en.name, w, cyc_wave, cyc_simd, ginst, tick_ghz);
            char buf[256];
            snprintf(buf, sizeof buf, "%s\"w%d\": {\"cycles_per_inst_per_wave\": %.4f, \"cycles_per_inst_per_simd\": %.4f, \"ginst_per_s_chip\": %.2f, \"launch_ms\": %.4f}",
                     wi ? ", " : "", w, cyc_wave, cyc_simd, ginst, ms);
            json += buf;
        }
        json += "}";
    }
    json += "}}\n";
    (void)simds;
    if (argc > 1) {
        FILE *f = fopen(argv[1], "w");
        if (f) { fputs(json.c_str(), f); fclose(f); }
    }
    return 0;
}
