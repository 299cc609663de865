// tools/ubench_issue.hip -- issue cost (cycles per wave64 instruction per SIMD) of the opcodes that make up the
// integer kernels of the proving path (Poseidon leaf hashing, NTT butterflies), with the operand shapes those kernels use.
//
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_issue.hip -o tools/ubench_issue && tools/ubench_issue > profiles/r02_ubench_issue_cost.json
//
// Method: every test is 32 instances of one instruction (16 independent dependency chains x 2) in a loop of 2048
// iterations, launched with enough 256-thread workgroups for 8 waves per SIMD on every CU (the occupancy of the leaf-hashing
// kernel: 62 VGPRs).  Two clocks are reported: wall time x the nominal 2.4 GHz (what a roofline against the spec clock
// needs) and the s_memtime shader-cycle counter read by wave 0 of every workgroup around its own loop, divided by the 8
// co-resident waves (independent of DVFS).  tools/isa_histogram.py multiplies these costs with a kernel's dynamic opcode
// histogram to get its issue ceiling.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#define ITERS 2048
#define CH 16
#define REP 2

enum {
    OP_ADD_U32, OP_MOV_B32, OP_XOR_B32, OP_AND_B32, OP_LSHLREV_B32, OP_LSHRREV_B32, OP_ASHRREV_I32, OP_SUB_U32,
    OP_ADD_CO_VCC, OP_ADDC_CO_VCC, OP_SUB_CO_VCC, OP_SUBBREV_CO_VCC, OP_ADD_CO_E64, OP_SUB64_PAIR, OP_ADD64_PAIR,
    OP_CNDMASK_SGPR, OP_CNDMASK_VCC_FIXED, OP_BITOP3, OP_ADD3, OP_ALIGNBIT, OP_PERM, OP_LSHL_ADD_U64, OP_LSHLREV_B64, OP_LSHRREV_B64,
    OP_MUL_LO, OP_MUL_HI, OP_MAD_U32_U24, OP_MAD64_VV, OP_MAD64_SV, OP_MAD64_CONST, OP_MAD64_SV_CARRYUSE, OP_CMP_LT_U64, OP_MOV_B64,
    OP_MIX_MAD_MOV, OP_MIX_MAD_ADD, OP_MAD_NOP, OP_READLANE, OP_COUNT
};

template <int OP>
__global__ __launch_bounds__(256) void k(uint64_t* out, uint64_t seed, unsigned long long* cyc) {
    uint32_t lane = threadIdx.x + blockIdx.x * blockDim.x;
    uint32_t a[CH], b[CH];
    uint64_t x[CH];
#pragma unroll
    for (int i = 0; i < CH; i++) { a[i] = (uint32_t)(seed * (lane + 1) + i); b[i] = a[i] ^ 0x9e3779b9u; x[i] = seed * (lane + 3) + i; }
    uint32_t m = (uint32_t)seed | 1u;
    uint64_t m64 = seed | 0x8000000000000001ULL;
    uint32_t sc = (uint32_t)__builtin_amdgcn_readfirstlane((int)(seed >> 3)) | 17u;  // wave-uniform multiplier in an SGPR
    uint64_t smask = __builtin_amdgcn_read_exec() ^ (seed << 7);                        // an SGPR pair used as a select mask
    asm volatile("" : "+s"(sc), "+s"(smask));
    unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int r = 0; r < REP; r++)
#pragma unroll
            for (int i = 0; i < CH; i++) {
                if (OP == OP_ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (OP == OP_SUB_U32) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (OP == OP_MOV_B32) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
                if (OP == OP_XOR_B32) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (OP == OP_AND_B32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (OP == OP_LSHLREV_B32) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[i]));
                if (OP == OP_LSHRREV_B32) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a[i]));
                if (OP == OP_ASHRREV_I32) asm volatile("v_ashrrev_i32 %0, 31, %1" : "=v"(a[i]) : "v"(b[i]));
                if (OP == OP_ADD_CO_VCC) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(m) : "vcc");
                if (OP == OP_ADDC_CO_VCC) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
                if (OP == OP_SUB_CO_VCC) asm volatile("v_sub_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(m) : "vcc");
                if (OP == OP_SUBBREV_CO_VCC) asm volatile("v_subbrev_co_u32 %0, vcc, 0, %0, vcc" : "+v"(a[i]) : : "vcc");
                if (OP == OP_ADD_CO_E64) { uint64_t c; asm volatile("v_add_co_u32_e64 %0, %1, %0, %2" : "+v"(a[i]), "=s"(c) : "v"(m)); }
                // a 64-bit subtract / add as the compiler emits it (two instructions): cost reported per instruction
                if (OP == OP_SUB64_PAIR) asm volatile("v_sub_co_u32 %0, vcc, %0, %2\n\tv_subbrev_co_u32 %1, vcc, 0, %1, vcc" : "+v"(a[i]), "+v"(b[i]) : "v"(m) : "vcc");
                if (OP == OP_ADD64_PAIR) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(a[i]), "+v"(b[i]) : "v"(m) : "vcc");
                if (OP == OP_CNDMASK_SGPR) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "s"(smask));
                if (OP == OP_CNDMASK_VCC_FIXED) asm volatile("v_cndmask_b32 %0, 0, -1, %1" : "=v"(a[i]) : "s"(smask));
                if (OP == OP_BITOP3) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x48" : "+v"(a[i]) : "v"(b[i]), "v"(m));
                if (OP == OP_ADD3) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m));
                if (OP == OP_ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[i]) : "v"(b[i]));
                if (OP == OP_PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(m));
                if (OP == OP_LSHL_ADD_U64) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(x[i]) : "v"(m64));
                if (OP == OP_LSHLREV_B64) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(x[i]));
                if (OP == OP_LSHRREV_B64) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(x[i]));
                if (OP == OP_MUL_LO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (OP == OP_MUL_HI) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (OP == OP_MAD_U32_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(m));
                // 32 x 32 + 64 multiply-add: VGPR x VGPR (64 x 64 product), VGPR x SGPR (MDS entries), VGPR x inline constant (reduction)
                if (OP == OP_MAD64_VV) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x[i]) : "v"(a[i]), "v"(m) : "vcc");
                if (OP == OP_MAD64_SV) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x[i]) : "v"(a[i]), "s"(sc) : "vcc");
                if (OP == OP_MAD64_CONST) asm volatile("v_mad_u64_u32 %0, vcc, %1, -1, %0" : "+v"(x[i]) : "v"(a[i]) : "vcc");
                // the carry-out idiom of gl_reduce128 / gl_mul_wide: mad, s_nop 1, cndmask on the mad's SGPR carry (cost per 3 "slots")
                if (OP == OP_MAD64_SV_CARRYUSE) { uint64_t c; asm volatile("v_mad_u64_u32 %0, %1, %2, -1, %0\n\ts_nop 1\n\tv_cndmask_b32 %3, 0, -1, %1" : "+v"(x[i]), "=&s"(c), "+v"(a[i]), "=v"(b[i])); }
                if (OP == OP_CMP_LT_U64) asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(x[i]), "v"(m64) : "vcc");
                if (OP == OP_MOV_B64) asm volatile("v_mov_b64 %0, %1" : "=v"(x[i]) : "v"(m64));
                // mixes: does a cheap VOP1/VOP2 hide beside a multiply-add?  (cost reported per instruction PAIR / triple)
                if (OP == OP_MIX_MAD_MOV) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mov_b32 %3, %1" : "+v"(x[i]), "+v"(a[i]) : "s"(sc), "v"(b[i]) : "vcc");
                if (OP == OP_MIX_MAD_ADD) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_add_u32 %3, %3, %1" : "+v"(x[i]) : "v"(a[i]), "s"(sc), "v"(b[i]) : "vcc");
                if (OP == OP_MAD_NOP) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\ts_nop 1" : "+v"(x[i]) : "v"(a[i]), "s"(sc) : "vcc");
                if (OP == OP_READLANE) { uint32_t s; asm volatile("v_readlane_b32 %0, %1, 0" : "=s"(s) : "v"(a[i])); }
            }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < CH; i++) acc ^= a[i] ^ x[i] ^ b[i];
    out[lane] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

struct result { const char* name; double wall_cycles, counter_cycles; int per; };

template <int OP> static result run(const char* name, int instr_per_instance, uint64_t* d, unsigned long long* dc, double ghz, int ncu) {
    const int blocks = ncu * 8, threads = 256;  // 8 workgroups x 4 waves per CU = 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 0x1234567ULL, dc); hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 3;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 0x1234567ULL, dc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), dc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += (double)v; avg /= blocks;
    double inst = (double)ITERS * CH * REP * instr_per_instance;          // per wave
    double wave_ops = (double)blocks * threads / 64 * inst;
    result r{name, ms * 1e-3 * ghz * 1e9 * (ncu * 4) / wave_ops, avg / (8.0 * inst), instr_per_instance};
    return r;
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    double ghz = p.clockRate / 1e6; int ncu = p.multiProcessorCount;
    uint64_t* d; hipMalloc(&d, (size_t)ncu * 8 * 256 * 8);
    unsigned long long* dc; hipMalloc(&dc, (size_t)ncu * 8 * 8);
    std::vector<result> rs;
#define RUN(OP, NAME, PER) rs.push_back(run<OP>(NAME, PER, d, dc, ghz, ncu))
    RUN(OP_ADD_U32, "v_add_u32", 1); RUN(OP_SUB_U32, "v_sub_u32", 1); RUN(OP_MOV_B32, "v_mov_b32", 1); RUN(OP_XOR_B32, "v_xor_b32", 1);
    RUN(OP_AND_B32, "v_and_b32", 1); RUN(OP_LSHLREV_B32, "v_lshlrev_b32", 1); RUN(OP_LSHRREV_B32, "v_lshrrev_b32", 1);
    RUN(OP_ASHRREV_I32, "v_ashrrev_i32", 1); RUN(OP_ADD_CO_VCC, "v_add_co_u32", 1); RUN(OP_ADDC_CO_VCC, "v_addc_co_u32", 1);
    RUN(OP_SUB_CO_VCC, "v_sub_co_u32", 1); RUN(OP_SUBBREV_CO_VCC, "v_subbrev_co_u32", 1); RUN(OP_ADD_CO_E64, "v_add_co_u32(sgpr carry)", 1);
    RUN(OP_SUB64_PAIR, "pair:v_sub_co_u32+v_subbrev_co_u32", 2); RUN(OP_ADD64_PAIR, "pair:v_add_co_u32+v_addc_co_u32", 2);
    RUN(OP_CNDMASK_SGPR, "v_cndmask_b32", 1); RUN(OP_CNDMASK_VCC_FIXED, "v_cndmask_b32(consts)", 1); RUN(OP_BITOP3, "v_bitop3_b32", 1);
    RUN(OP_ADD3, "v_add3_u32", 1); RUN(OP_ALIGNBIT, "v_alignbit_b32", 1); RUN(OP_PERM, "v_perm_b32", 1);
    RUN(OP_LSHL_ADD_U64, "v_lshl_add_u64", 1); RUN(OP_LSHLREV_B64, "v_lshlrev_b64", 1); RUN(OP_LSHRREV_B64, "v_lshrrev_b64", 1);
    RUN(OP_MUL_LO, "v_mul_lo_u32", 1); RUN(OP_MUL_HI, "v_mul_hi_u32", 1); RUN(OP_MAD_U32_U24, "v_mad_u32_u24", 1);
    RUN(OP_MAD64_VV, "v_mad_u64_u32(vgpr x vgpr)", 1); RUN(OP_MAD64_SV, "v_mad_u64_u32(vgpr x sgpr)", 1); RUN(OP_MAD64_CONST, "v_mad_u64_u32(vgpr x const)", 1);
    RUN(OP_MAD64_SV_CARRYUSE, "triple:v_mad_u64_u32+s_nop+v_cndmask_b32", 3); RUN(OP_CMP_LT_U64, "v_cmp_lt_u64", 1); RUN(OP_MOV_B64, "v_mov_b64", 1);
    RUN(OP_MIX_MAD_MOV, "pair:v_mad_u64_u32+v_mov_b32", 2); RUN(OP_MIX_MAD_ADD, "pair:v_mad_u64_u32+v_add_u32", 2);
    RUN(OP_MAD_NOP, "pair:v_mad_u64_u32+s_nop1", 2); RUN(OP_READLANE, "v_readlane_b32", 1);
    // JSON: the cost table isa_histogram.py reads uses the counter-based cycles (DVFS independent); v_mad_u64_u32 = the vgpr x sgpr form
    printf("{\n \"device\": \"%s\", \"cus\": %d, \"nominal_ghz\": %.3f, \"waves_per_simd\": 8,\n", p.gcnArchName, ncu, ghz);
    printf(" \"method\": \"32 instances (16 chains x 2) per loop iteration, 2048 iterations, 8 waves per SIMD; counter = s_memtime cycles of one wave / (8 waves x its instructions); wall = event time x nominal clock x SIMDs / wave-instructions\",\n");
    printf(" \"raw\": [\n");
    for (size_t i = 0; i < rs.size(); i++)
        printf("  {\"test\": \"%s\", \"instructions_per_instance\": %d, \"cycles_counter\": %.3f, \"cycles_wall_nominal\": %.3f}%s\n", rs[i].name, rs[i].per,
               rs[i].counter_cycles, rs[i].wall_cycles, i + 1 < rs.size() ? "," : "");
    printf(" ],\n \"cycles_per_wave_instr\": {\n");
    bool first = true;
    for (auto& r : rs) {
        std::string n = r.name;
        if (n.rfind("pair:", 0) == 0 || n.rfind("triple:", 0) == 0) continue;
        if (n == "v_mad_u64_u32(vgpr x vgpr)" || n == "v_mad_u64_u32(vgpr x const)" || n == "v_add_co_u32(sgpr carry)" || n == "v_cndmask_b32(consts)") continue;
        size_t paren = n.find('(');
        if (paren != std::string::npos) n = n.substr(0, paren);
        printf("%s  \"%s\": %.3f", first ? "" : ",\n", n.c_str(), r.counter_cycles);
        first = false;
    }
    printf(",\n  \"_default_vop3\": 4.4\n }\n}\n");
    return 0;
}
