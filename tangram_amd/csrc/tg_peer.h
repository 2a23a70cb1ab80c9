// tg_peer.h -- the peer-memory transport of the spot-sharded step: mailbox granules, the exchange kernel.
// Included by tg_kernels.h.
#pragma once
// ----------------------------------------------------------------------------------------------
// Peer-memory exchange: the third tg_comm transport (tg_capi.hip: tg_comm_peer_create / _connect).
//   The three per-step exchanges of a spot shard are 8 KB - 1 MB vectors whose cost on a collective library is its fixed latency
//   (a ring all-reduce on 8 ranks is 14 dependent hops).  Here an exchange is ONE kernel per rank and ONE hop: every rank owns a
//   MAILBOX in its own HBM that every peer has mapped (hipIpc between processes of a node: xGMI stores; plain pointers between
//   shards inside one process).  The mailbox holds, per generation slot (2) and source rank, the vector as 8-byte GRANULES
//   {float value, sequence number of the exchange}, each written by ONE naturally aligned write-through store (system-scope relaxed
//   atomic: sc0 sc1) -- value and tag arrive together or not at all, so there is no flag, no fence and no barrier
//   (MI355X_MICROARCH.md, "handoff-1to1": tagged granules cost half of payload + flag).  A thread
//     1. reads its elements of this rank's vector and stores their granules into slot [seq & 1][this rank] of EVERY rank's mailbox
//        (its own included), peers visited from rank + 1 on;
//     2. polls the granules of the same elements from every rank in its OWN mailbox until their tag is this exchange's sequence
//        number (bounded: a peer that never arrives costs TG_PEER_TIMEOUT_MS, raises the error word, and the kernel ends);
//     3. all-reduce: adds the world values in RANK ORDER (every rank adds the same floats in the same order: bit-identical on every
//        rank, and equal to the callback transport's rank-order sum); all-gather: copies them out.
//   Elements are independent: no workgroup or grid barrier.  Two generation slots suffice: a rank leaves exchange g only after every
//   peer has pushed g, and a peer pushes g + 1 only after its own kernel of g has finished reading, so nobody writes generation g + 2
//   into a slot somebody still reads generation g from.  Sequence numbers only grow (the mailbox starts zeroed; the first is 1).
// ----------------------------------------------------------------------------------------------
#define TG_PEER_MAX 16
#define TG_PEER_CHUNK 512               // floats per workgroup (two per thread: see tg_peer_exchange)
#define TG_PEER_HDR 256                 // bytes in front of the granules: [0] error word (1: a poll timed out)
#ifdef TG_SIM
#include <chrono>
#include <sched.h>
TG_DEV void tg_sys_store_u64(unsigned long long* p, unsigned long long v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
TG_DEV unsigned long long tg_sys_load_u64(const unsigned long long* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
TG_DEV void tg_sys_store_u32(unsigned* p, unsigned v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
TG_DEV unsigned tg_sys_load_u32(const unsigned* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
TG_DEV unsigned long long tg_wall_ticks() {      // 100 MHz like wall_clock64()
    return (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() / 10;
}
TG_DEV void tg_poll_pause() { sched_yield(); }
#else
TG_DEV void tg_sys_store_u64(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
TG_DEV unsigned long long tg_sys_load_u64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
TG_DEV void tg_sys_store_u32(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
TG_DEV unsigned tg_sys_load_u32(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
TG_DEV unsigned long long tg_wall_ticks() { return wall_clock64(); }
TG_DEV void tg_poll_pause() { __builtin_amdgcn_s_sleep(2); }
#endif
struct TgPeerArgs {
    unsigned char* box[TG_PEER_MAX];    // every rank's mailbox as mapped by THIS rank; box[rank] is its own
    int world, rank;
    unsigned long long cap;             // granules of one (slot, rank) region
    int slot; unsigned seq;
    const float* send; float* recv;     // all-reduce: in place (send == recv)
    unsigned long long n;               // floats (per rank)
    int gather; unsigned long long ld;  // gather: recv[r * ld + i]
    unsigned long long timeout_ticks;   // bound of a poll in 10-ns ticks
};
TG_HD size_t tg_peer_box_bytes(int world, size_t cap, size_t step_cap) { return TG_PEER_HDR + (size_t)2 * world * (cap + step_cap) * 8; }
// the value of granule *g once its tag is `seq` (0.f after a time-out, with the error word raised).  A raised error word ends EVERY wait
// of this rank at its next look (a lost peer costs one bounded wait per rank, then the run drains quickly and tg_comm_peer_status says why).
TG_DEV float tg_peer_take(const unsigned long long* g, unsigned seq, unsigned long long timeout, unsigned* err) {
    unsigned long long x = tg_sys_load_u64(g);
    if ((unsigned)(x >> 32) != seq) {
        const unsigned long long t0 = tg_wall_ticks();
        do {
            tg_poll_pause();
            x = tg_sys_load_u64(g);
            if ((unsigned)(x >> 32) == seq) break;
            if (tg_sys_load_u32(err) != 0u || tg_wall_ticks() - t0 > timeout) { tg_sys_store_u32(err, 1u); return 0.f; }
        } while (true);
    }
    return __builtin_bit_cast(float, (unsigned)x);
}
// a granule that was already requested (x): its value if the tag is there, else the patient path above
TG_DEV float tg_peer_value(unsigned long long x, const unsigned long long* g, unsigned seq, unsigned long long timeout, unsigned* err) {
    return ((unsigned)(x >> 32) == seq) ? __builtin_bit_cast(float, (unsigned)x) : tg_peer_take(g, seq, timeout, err);
}
// Round 6: every granule a thread needs is REQUESTED before the first one is looked at -- two elements x up to eight ranks in flight per
// thread.  Round 5 took them one after the other (eight elements per thread, one dependent round trip to the fine-grained mailbox each,
// ~2 us apiece: the 17 - 19 us per exchange of profiles/r05/final/shard.json were eight such trips, not the fabric).
TG_KERNEL void TG_LAUNCH_BOUNDS(256) tg_peer_exchange(TgPeerArgs a) {
    constexpr int NJ = TG_PEER_CHUNK / 256, RB = 8;
    const int t = threadIdx.x;
    const size_t lo = (size_t)blockIdx.x * TG_PEER_CHUNK;
    const size_t region = ((size_t)a.slot * a.world + a.rank) * a.cap;         // my region in anybody's mailbox
    const unsigned long long tag = (unsigned long long)a.seq << 32;
    // 1. push
    float mine[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) { const size_t i = lo + t + 256 * (size_t)j; mine[j] = i < a.n ? a.send[i] : 0.f; }
    for (int p = 0; p < a.world; ++p) {
        unsigned long long* dst = (unsigned long long*)(a.box[(a.rank + 1 + p) % a.world] + TG_PEER_HDR) + region;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const size_t i = lo + t + 256 * (size_t)j;
            if (i < a.n) tg_sys_store_u64(dst + i, tag | (unsigned long long)__builtin_bit_cast(unsigned, mine[j]));
        }
    }
    // 2. + 3. take every rank's granules of my elements out of MY mailbox, RB ranks x NJ elements requested at a time
    const unsigned long long* in = (const unsigned long long*)(a.box[a.rank] + TG_PEER_HDR) + (size_t)a.slot * a.world * a.cap;
    unsigned* err = (unsigned*)a.box[a.rank];
    float s[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) s[j] = 0.f;
    for (int r0 = 0; r0 < a.world; r0 += RB) {
        unsigned long long x[NJ][RB];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const size_t i = lo + t + 256 * (size_t)j;
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                const int r = (r0 + q < a.world) ? r0 + q : a.world - 1;
                x[j][q] = (i < a.n) ? tg_sys_load_u64(in + (size_t)r * a.cap + i) : 0ull;
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const size_t i = lo + t + 256 * (size_t)j;
            if (i >= a.n) continue;
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                const int r = r0 + q;
                if (r >= a.world) continue;
                const float v = tg_peer_value(x[j][q], in + (size_t)r * a.cap + i, a.seq, a.timeout_ticks, err);
                if (a.gather) a.recv[(size_t)r * a.ld + i] = v;
                else s[j] = (r == 0) ? v : s[j] + v;                    // rank order
            }
        }
    }
    if (!a.gather) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) { const size_t i = lo + t + 256 * (size_t)j; if (i < a.n) a.recv[i] = s[j]; }
    }
}

// ----------------------------------------------------------------------------------------------
// Round 6: the three exchanges of a sharded step INSIDE the kernels around them ("a persistent peer-memory kernel instead of
// three RCCL launches").  Behind the generic area every mailbox has a STEP AREA of [2 slots][world][step_cap] granules with three fixed
// regions per (slot, source rank): E2 per-gene statistics [2][Kp], E3 row sums [TGP1_N][C], E1 row pairs [2][C] + TG_PAIR_TAIL.  A kernel
// that owns a value pushes its granule into that place of EVERY mailbox (tg_link_push) as soon as it has it; the kernel that needs the
// sum over ranks reads the world granules of that place out of its OWN mailbox (tg_link_sum / tg_link_get: requested together, summed
// in RANK ORDER).  Tags are the step's sequence number; slot = seq & 1.  Why two slots suffice with pushes and polls spread over
// kernels: a rank writes generation g + 2 of a place only after it has passed every exchange of generation g + 1, which needs the
// g + 1 pushes of every peer, which a peer issues only after ITS kernels of generation g (readers included) have finished.
// ----------------------------------------------------------------------------------------------
struct TgPeerLink {
    unsigned char* const* box;          // [world] every rank's mailbox as mapped by THIS rank, a table in ordinary device memory (an array inside
                                        // the kernel arguments, indexed by a run-time rank, is copied to scratch by the compiler: measured)
    int world, rank;                    // world == 0: no link (the exchange happens outside the kernel)
    unsigned long long base;            // granule offset of the step area inside a mailbox
    unsigned long long cap;             // granules per (slot, source rank) of the step area
    unsigned long long timeout_ticks;
    unsigned seq; int slot;
    unsigned long long e2, e3, e1;      // granule offsets of the three regions inside a (slot, rank) block
};
TG_DEV unsigned long long* tg_link_dst(const TgPeerLink& k, int to_rank, size_t idx) {          // MY granule `idx` in rank to_rank's mailbox
    return (unsigned long long*)(k.box[to_rank] + TG_PEER_HDR) + k.base + ((size_t)k.slot * k.world + k.rank) * k.cap + idx;
}
TG_DEV const unsigned long long* tg_link_src(const TgPeerLink& k, int from_rank, size_t idx) {  // rank from_rank's granule `idx` in MY mailbox
    return (const unsigned long long*)(k.box[k.rank] + TG_PEER_HDR) + k.base + ((size_t)k.slot * k.world + from_rank) * k.cap + idx;
}
TG_DEV void tg_link_push_to(const TgPeerLink& k, int to_rank, size_t idx, float v) {
    tg_sys_store_u64(tg_link_dst(k, to_rank, idx), ((unsigned long long)k.seq << 32) | (unsigned long long)__builtin_bit_cast(unsigned, v));
}
TG_DEV void tg_link_push(const TgPeerLink& k, size_t idx, float v) {
    const unsigned long long g = ((unsigned long long)k.seq << 32) | (unsigned long long)__builtin_bit_cast(unsigned, v);
    for (int p = 0; p < k.world; ++p) tg_sys_store_u64(tg_link_dst(k, (k.rank + 1 + p) % k.world, idx), g);
}
// granule `idx` of ranks r0 .. r0 + 7 (clamped to the world), all requested before the first is looked at: values in v[0 .. 7]
TG_DEV void tg_link_get8(const TgPeerLink& k, size_t idx, int r0, float (&v)[8]) {
    unsigned* err = (unsigned*)k.box[k.rank];
    unsigned long long x[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) x[q] = tg_sys_load_u64(tg_link_src(k, (r0 + q < k.world) ? r0 + q : k.world - 1, idx));
#pragma unroll
    for (int q = 0; q < 8; ++q)
        v[q] = (r0 + q < k.world) ? tg_peer_value(x[q], tg_link_src(k, r0 + q, idx), k.seq, k.timeout_ticks, err) : 0.f;
}
// sum over ranks of granule `idx`, in rank order
TG_DEV float tg_link_sum(const TgPeerLink& k, size_t idx) {
    float s = 0.f;
    for (int r0 = 0; r0 < k.world; r0 += 8) {
        float v[8];
        tg_link_get8(k, idx, r0, v);
#pragma unroll
        for (int q = 0; q < 8; ++q) if (r0 + q < k.world) s = (r0 + q == 0) ? v[q] : s + v[q];
    }
    return s;
}
// two places at once (their granules all requested together): the values of ranks r0 .. r0 + 7 of both
TG_DEV void tg_link_get8x2(const TgPeerLink& k, size_t ia, size_t ib, int r0, float (&va)[8], float (&vb)[8]) {
    unsigned* err = (unsigned*)k.box[k.rank];
    unsigned long long xa[8], xb[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int r = (r0 + q < k.world) ? r0 + q : k.world - 1;
        xa[q] = tg_sys_load_u64(tg_link_src(k, r, ia));
        xb[q] = tg_sys_load_u64(tg_link_src(k, r, ib));
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const bool on = r0 + q < k.world;
        va[q] = on ? tg_peer_value(xa[q], tg_link_src(k, r0 + q, ia), k.seq, k.timeout_ticks, err) : 0.f;
        vb[q] = on ? tg_peer_value(xb[q], tg_link_src(k, r0 + q, ib), k.seq, k.timeout_ticks, err) : 0.f;
    }
}
TG_DEV void tg_link_sum2(const TgPeerLink& k, size_t ia, size_t ib, float& sa, float& sb) {
    sa = sb = 0.f;
    for (int r0 = 0; r0 < k.world; r0 += 8) {
        float va[8], vb[8];
        tg_link_get8x2(k, ia, ib, r0, va, vb);
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (r0 + q < k.world) { sa = (r0 + q == 0) ? va[q] : sa + va[q]; sb = (r0 + q == 0) ? vb[q] : sb + vb[q]; }
    }
}
