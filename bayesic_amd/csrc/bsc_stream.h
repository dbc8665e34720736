// Shared by the persistent kernels (csrc/bsc_gemm.hip, csrc/bsc_lda.hip): how a grid of at most two
// workgroups per CU walks a list of tiles of `n_kt` units each -- whole rounds of tiles dealt
// round-robin, then the left-over tiles split along their units among all workgroups (or one more
// partly empty round) -- and the scalar-unit hygiene that loop needs on gfx950.
//
// The argument struct G must carry: int n_kt, sk_q, sk_r, sk_stream, n_wg, rounds, tail_tiles.
#pragma once
#include "bsc_common.h"

// A quotient of wave-uniform values, said to be uniform: the division itself runs on the vector
// unit, and everything computed from an unmarked result -- tile coordinates, descriptors, loop
// conditions -- would follow it there (exec-masked branches, readfirstlane loops around each DMA).
__device__ __forceinline__ unsigned stream_udiv(unsigned a, unsigned b) {
    return (unsigned)__builtin_amdgcn_readfirstlane((int)(a / b));
}

// The kernel's arguments, re-read from the kernarg segment where they are needed: a persistent
// kernel touches most of its arguments only at tile boundaries, and held in scalar registers across
// the k-loop they cost it ~120 SGPR spills (v_writelane / v_readlane in the loop: vector
// instructions, which take their cycles from the matrix pipe).  The empty asm keeps the loads
// from being hoisted back out.
template <class G>
using stream_args_cptr = __attribute__((address_space(4))) const G*;
template <class G>
__device__ __forceinline__ stream_args_cptr<G> stream_cold_args() {       // G = the kernel's first and only argument
    stream_args_cptr<G> p = (stream_args_cptr<G>)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

// first tail unit of workgroup w (w = n_wg: the end of the list)
template <class GP>
__device__ __forceinline__ int stream_first_unit(GP g, int w) {
    if (g->sk_stream) return w * g->sk_q + (w < g->sk_r ? w : g->sk_r);
    return (w < g->tail_tiles ? w : g->tail_tiles) * g->n_kt;
}

// Where a workgroup's run stands: k-tile `kt` of tile `t`, `left` more units of that tile to go.
struct StreamCursor {
    int t, round, kt, left, tail_left;
    template <class GP>
    __device__ __forceinline__ void segment(GP g, int w, int tail_u0) {
        const int rounds = g->rounds, n_kt = g->n_kt;
        if (round < rounds) {
            t = round * g->n_wg + w;
            kt = 0;
            left = n_kt;
        } else {
            if (round == rounds) {
                const int tt = (int)stream_udiv((unsigned)tail_u0, (unsigned)n_kt);
                t = rounds * g->n_wg + tt;
                kt = tail_u0 - tt * n_kt;
            } else {
                ++t;
                kt = 0;
            }
            left = n_kt - kt < tail_left ? n_kt - kt : tail_left;
            tail_left -= left;
        }
    }
    template <class GP>
    __device__ __forceinline__ void begin(GP g, int w, int tail_u0, int tail_cnt) {
        round = 0;
        tail_left = tail_cnt;
        t = 0; kt = 0; left = 0;
        segment(g, w, tail_u0);
    }
    // one unit on; true when that was the tile's last unit in this run (the caller then moves
    // to the next segment)
    __device__ __forceinline__ bool step() {
        ++kt;
        return --left == 0;
    }
};

__device__ __forceinline__ const float* stream_uniform_ptr(const float* p) {
    const uint64_t v = (uint64_t)(uintptr_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const float*)(uintptr_t)(((uint64_t)hi << 32) | lo);
}


// Host side: the schedule of `tiles` tiles of `n_kt` units on `slots` resident workgroups.  The tail
// is split along the units when that saves more than the pass over the partial results costs
// (about `piece_cost` units' time), else it is one more -- partly empty -- round.
template <class G>
inline void stream_plan(G& s, int64_t tiles, int n_kt, int64_t slots, int piece_cost = 6) {
    s.n_kt = n_kt;
    const int64_t left = tiles % slots;
    s.sk_stream = left > 0 && (int64_t)n_kt * (slots - left) >= piece_cost * slots;
    if (s.sk_stream) {
        const int64_t tail_units = left * n_kt;            // < 2^31: the caller bounds tiles and n_kt
        s.rounds = (int)(tiles / slots);
        s.n_wg = (int)(s.rounds > 0 || tail_units >= slots ? slots : tail_units);
        s.tail_tiles = (int)left;
        s.sk_q = (int)(tail_units / s.n_wg);
        s.sk_r = (int)(tail_units % s.n_wg);
    } else {
        s.n_wg = (int)(tiles < slots ? tiles : slots);
        s.rounds = (int)(tiles / s.n_wg);
        s.tail_tiles = (int)(tiles % s.n_wg);
        s.sk_q = 0;
        s.sk_r = 0;
    }
}
// whether a fix-up pass over partial tiles is needed
template <class G>
inline bool stream_has_pieces(const G& s) {
    return s.sk_stream && s.n_wg > 1 && !(s.sk_r == 0 && s.sk_q % s.n_kt == 0);
}
