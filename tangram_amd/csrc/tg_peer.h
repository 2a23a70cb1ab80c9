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
#define TG_PEER_CHUNK 2048              // floats per workgroup
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
TG_HD size_t tg_peer_box_bytes(int world, size_t cap) { return TG_PEER_HDR + (size_t)2 * world * cap * 8; }
// the value of granule *g once its tag is `seq` (0.f after a time-out, with the error word raised)
TG_DEV float tg_peer_take(const unsigned long long* g, unsigned seq, unsigned long long timeout, unsigned* err) {
    unsigned long long x = tg_sys_load_u64(g);
    if ((unsigned)(x >> 32) != seq) {
        const unsigned long long t0 = tg_wall_ticks();
        do {
            tg_poll_pause();
            x = tg_sys_load_u64(g);
            if ((unsigned)(x >> 32) == seq) break;
            if (tg_wall_ticks() - t0 > timeout) { tg_sys_store_u32(err, 1u); return 0.f; }
        } while (true);
    }
    return __builtin_bit_cast(float, (unsigned)x);
}
TG_KERNEL void TG_LAUNCH_BOUNDS(256) tg_peer_exchange(TgPeerArgs a) {
    constexpr int NJ = TG_PEER_CHUNK / 256;
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
    // 2. + 3. take every rank's granules of my elements out of MY mailbox.  Once a poll has timed out (error word raised) a peer is
    // gone: later exchanges do not wait again -- the run ends quickly with garbage and tg_comm_peer_status says why.
    const unsigned long long* in = (const unsigned long long*)(a.box[a.rank] + TG_PEER_HDR) + (size_t)a.slot * a.world * a.cap;
    unsigned* err = (unsigned*)a.box[a.rank];
    if (tg_sys_load_u32(err) != 0u) a.timeout_ticks = 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const size_t i = lo + t + 256 * (size_t)j;
        if (i >= a.n) continue;
        if (a.gather) {
            for (int r = 0; r < a.world; ++r) a.recv[(size_t)r * a.ld + i] = tg_peer_take(in + (size_t)r * a.cap + i, a.seq, a.timeout_ticks, err);
        } else {
            float s = tg_peer_take(in + i, a.seq, a.timeout_ticks, err);
            for (int r = 1; r < a.world; ++r) s += tg_peer_take(in + (size_t)r * a.cap + i, a.seq, a.timeout_ticks, err);
            a.recv[i] = s;
        }
    }
}
