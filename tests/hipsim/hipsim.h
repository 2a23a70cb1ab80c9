// hipsim.h -- TEST INFRASTRUCTURE ONLY (lives under tests/, never loaded by the product path).
//
// A tiny single-process emulator of the HIP execution model, just large enough to run the
// kernels of tangram_amd/csrc on the CPU of the authoring container (which has no GPU):
//   * one fiber (own stack, hand-written context switch) per work-item, one workgroup at a time, wavefront = 64 lanes;
//   * __syncthreads(), wave shuffles and the gfx950 MFMA builtins used by the kernels, with the
//     lane->element layouts of /opt/skills/guides/cdna_hip_programming.md section 3;
//   * "device memory" is host memory; a launch runs synchronously.
// It exists so that index arithmetic, masking, barrier placement and the host-side launch
// sequence can be checked against the oracle with `pytest -m "not gpu"`.  It says nothing about
// speed, and data races are only visible as far as the deterministic fiber order exposes them.
#pragma once
#if defined(__x86_64__)
#define HIPSIM_ASM_SWITCH 1          // hand-written context switch: swapcontext() costs two sigprocmask syscalls per switch
#else
#define HIPSIM_ASM_SWITCH 0
#include <ucontext.h>
#endif

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <vector>

#if HIPSIM_ASM_SWITCH
struct hipsim_ctx { void* sp; };
extern "C" __attribute__((visibility("hidden"))) void hipsim_switch(hipsim_ctx* from, hipsim_ctx* to);
// saves the callee-saved registers of the SysV ABI on the current stack, swaps stack pointers, restores, returns
asm(".text\n"
    ".type hipsim_switch,@function\n"
    "hipsim_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n"
    "  movq (%rsi), %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n"
    ".size hipsim_switch, .-hipsim_switch\n");
#endif

namespace hipsim {

struct uint3s { unsigned x, y, z; };

#if HIPSIM_ASM_SWITCH
typedef hipsim_ctx ctx_t;
inline void ctx_switch(ctx_t* from, ctx_t* to) { hipsim_switch(from, to); }
inline void ctx_make(ctx_t* c, char* stack, size_t size, void (*entry)()) {
    uintptr_t top = ((uintptr_t)stack + size) & ~(uintptr_t)15;
    void** sp = (void**)top;
    *--sp = nullptr;                  // fake return address of `entry` (it never returns)
    *--sp = (void*)entry;             // popped by the `ret` of the first switch
    for (int i = 0; i < 6; ++i) *--sp = nullptr;   // rbp rbx r12 r13 r14 r15
    c->sp = sp;
}
#else
typedef ucontext_t ctx_t;
inline void ctx_switch(ctx_t* from, ctx_t* to) { swapcontext(from, to); }
inline void ctx_make(ctx_t* c, char* stack, size_t size, void (*entry)()) {
    getcontext(c);
    c->uc_stack.ss_sp = stack;
    c->uc_stack.ss_size = size;
    c->uc_link = nullptr;
    makecontext(c, entry, 0);
}
#endif

struct Fiber {
    ctx_t ctx;
    char* stack = nullptr;
    uint3s tid{};
    int state = 0;            // 0 runnable, 1 waiting at block barrier, 2 done
    unsigned long block_gen = 0;
};

struct WaveSlot {             // scratch for wave-collective operations
    alignas(16) unsigned char a[64][16];
    alignas(16) unsigned char b[64][16];
    unsigned long long u64[64];
    int arrived = 0;
    unsigned long gen = 0;
};

struct Machine {
    uint3s gridDim{}, blockDim{}, blockIdx{};
    std::vector<Fiber> fibers;
    std::vector<WaveSlot> waves;
    ctx_t sched;
    Fiber* cur = nullptr;
    int cur_index = 0;
    int barrier_arrived = 0;
    unsigned long barrier_gen = 0;
    std::function<void()> body;
    alignas(16) unsigned char lds[160 * 1024];
};

inline Machine& M() { static Machine m; return m; }

inline void yield_to_sched() { Machine& m = M(); ctx_switch(&m.cur->ctx, &m.sched); }

inline void fiber_entry() {
    Machine& m = M();
    m.body();
    m.cur->state = 2;
    ctx_switch(&m.cur->ctx, &m.sched);
    abort();                           // a finished fiber is never resumed
}

constexpr size_t kStack = 256 * 1024;

template <class F>
void run_block(F&& f) {
    Machine& m = M();
    const int n = int(m.blockDim.x * m.blockDim.y * m.blockDim.z);
    if ((int)m.fibers.size() < n) {
        size_t old = m.fibers.size();
        m.fibers.resize(n);
        for (size_t i = old; i < (size_t)n; ++i) m.fibers[i].stack = (char*)malloc(kStack);
    }
    const int nwaves = (n + 63) / 64;
    m.waves.assign(nwaves, WaveSlot());
    m.body = f;
    m.barrier_arrived = 0;
    for (int i = 0; i < n; ++i) {
        Fiber& fb = m.fibers[i];
        fb.state = 0;
        fb.tid.x = i % m.blockDim.x;
        fb.tid.y = (i / m.blockDim.x) % m.blockDim.y;
        fb.tid.z = i / (m.blockDim.x * m.blockDim.y);
        ctx_make(&fb.ctx, fb.stack, kStack, fiber_entry);
    }
    int done = 0;
    while (done < n) {
        bool progressed = false;
        for (int w = 0; w < nwaves; ++w) {
            // run this wave until every lane is blocked at the block barrier or finished
            for (;;) {
                bool any = false;
                for (int l = 0; l < 64; ++l) {
                    int i = w * 64 + l;
                    if (i >= n) break;
                    Fiber& fb = m.fibers[i];
                    if (fb.state == 1 && fb.block_gen != m.barrier_gen) fb.state = 0;
                    if (fb.state != 0) continue;
                    any = true;
                    m.cur = &fb;
                    m.cur_index = i;
                    ctx_switch(&m.sched, &fb.ctx);
                    if (fb.state == 2) ++done;
                    progressed = true;
                }
                if (!any) break;
            }
        }
        if (!progressed) {
            fprintf(stderr, "hipsim: deadlock (divergent barrier?) in block (%u,%u)\n", m.blockIdx.x, m.blockIdx.y);
            abort();
        }
    }
}

inline void block_barrier() {
    Machine& m = M();
    const int n = int(m.blockDim.x * m.blockDim.y * m.blockDim.z);
    Fiber* me = m.cur;
    if (++m.barrier_arrived == n) {
        m.barrier_arrived = 0;
        ++m.barrier_gen;          // releases everybody (they compare generations)
        return;
    }
    me->block_gen = m.barrier_gen;
    me->state = 1;
    yield_to_sched();
}

inline int lane_id() { return M().cur_index & 63; }
inline WaveSlot& my_wave() { return M().waves[M().cur_index >> 6]; }
inline int wave_width() {
    Machine& m = M();
    const int n = int(m.blockDim.x * m.blockDim.y * m.blockDim.z);
    int w = m.cur_index >> 6;
    int rem = n - w * 64;
    return rem < 64 ? rem : 64;
}

inline void wave_barrier() {
    WaveSlot& ws = my_wave();
    const int n = wave_width();
    unsigned long g = ws.gen;
    if (++ws.arrived == n) { ws.arrived = 0; ++ws.gen; return; }
    while (ws.gen == g) yield_to_sched();
}

template <class T>
inline T shfl_idx(T v, int src) {
    static_assert(sizeof(T) <= 8, "shfl type too wide");
    WaveSlot& ws = my_wave();
    unsigned long long raw = 0;
    memcpy(&raw, &v, sizeof(T));
    ws.u64[lane_id()] = raw;
    wave_barrier();
    unsigned long long r = ws.u64[src & 63];
    wave_barrier();
    T out;
    memcpy(&out, &r, sizeof(T));
    return out;
}

inline float bf16_to_f32(uint16_t h) { uint32_t u = uint32_t(h) << 16; float f; memcpy(&f, &u, 4); return f; }
inline uint16_t f32_to_bf16(float f) {   // round to nearest even (v_cvt_pk_bf16_f32)
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return uint16_t((u >> 16) | 0x40);
    uint32_t lsb = (u >> 16) & 1u;
    u += 0x7fffu + lsb;
    return uint16_t(u >> 16);
}

// D = A*B + C, 16x16 tile, 64 lanes.  A/B fragments are 16 bytes per lane.
//   bf16 16x16x32: lane l holds A[row=l&15][k=8*(l>>4)+j], B[k=8*(l>>4)+j][col=l&15], j<8
//   f32  16x16x4 : lane l holds A[row=l&15][k=l>>4], B[k=l>>4][col=l&15]   (first 4 bytes)
//   C/D          : lane l, reg r -> row = 4*(l>>4)+r, col = l&15
inline void mfma16_bf16(const void* a, const void* b, float* c4) {
    WaveSlot& ws = my_wave();
    const int l = lane_id();
    memcpy(ws.a[l], a, 16);
    memcpy(ws.b[l], b, 16);
    wave_barrier();
    const int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (l >> 4) + r;
        double acc = c4[r];
        for (int k = 0; k < 32; ++k) {
            uint16_t ha, hb;
            memcpy(&ha, &ws.a[row + 16 * (k >> 3)][2 * (k & 7)], 2);
            memcpy(&hb, &ws.b[col + 16 * (k >> 3)][2 * (k & 7)], 2);
            acc += double(bf16_to_f32(ha)) * double(bf16_to_f32(hb));
        }
        c4[r] = float(acc);
    }
    wave_barrier();
}

inline void mfma16_f32(float a, float b, float* c4) {
    WaveSlot& ws = my_wave();
    const int l = lane_id();
    memcpy(ws.a[l], &a, 4);
    memcpy(ws.b[l], &b, 4);
    wave_barrier();
    const int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (l >> 4) + r;
        float acc = c4[r];
        for (int k = 0; k < 4; ++k) {
            float fa, fb;
            memcpy(&fa, ws.a[row + 16 * k], 4);
            memcpy(&fb, ws.b[col + 16 * k], 4);
            acc = fmaf(fa, fb, acc);      // hardware: k-ordered fmaf chain
        }
        c4[r] = acc;
    }
    wave_barrier();
}

inline std::mutex& launch_mutex() { static std::mutex m; return m; }   // (not inside the template: one mutex for ALL kernels)

template <class F>
void launch(uint3s grid, uint3s block, F&& f) {
    std::lock_guard<std::mutex> lock(launch_mutex());   // ONE emulated machine per process: launches from several host threads queue up
    Machine& m = M();
    m.gridDim = grid;
    m.blockDim = block;
    for (unsigned z = 0; z < grid.z; ++z)
        for (unsigned y = 0; y < grid.y; ++y)
            for (unsigned x = 0; x < grid.x; ++x) {
                m.blockIdx = {x, y, z};
                run_block(f);
            }
}

}  // namespace hipsim
