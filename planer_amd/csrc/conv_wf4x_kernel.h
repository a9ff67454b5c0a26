// Fully fused Winograd F(4x4,3x3), second form: the filter never enters LDS -- included by conv_winograd.hip after
// conv_wf4_kernel.h (shares Wf4Args, the transforms and the row tail with it).
//
// Knock-out timing of conv_wf4_kernel (tools/wf4_knock.sh) showed what a K step costs beside its MFMAs (1.00 us): fragment
// reads out of LDS 0.36 us, the patch transform 0.38 us, both LDS-DMA streams together 0.13 us -- the step is bound by LDS
// traffic (280 KB per step on a 128 B/clk array = 2190 of the step's 2304 matrix cycles), not by what it pulls from L2.
// This form moves 110 KB per step through LDS instead:
//   * 12 waves (three per SIMD, 168 registers).  Wave (h, b) owns output channels [32 h, 32 h + 32) x all 32 tiles x the six
//     frequencies (a = 0..5, b) of ONE column of the 6 x 6 frequency tile: 2 x 2 accumulator blocks of
//     v_mfma_f32_16x16x4_f32 x 6 = 96 registers.  Every filter value is needed by exactly one wave, so the A operand goes
//     global -> registers (three 16-byte loads per lane and step from a filter laid out [wave][kk][i][cb][a], requested a
//     step ahead as the registers retire) and LDS holds no filter: -36.9 KB of LDS-DMA writes and -73.7 KB of reads per step.
//   * V[b][kk][a pair][tile][2]: a wave reads its column as six conflict-free 8-byte reads per lane; each V value is read by
//     two waves (the two channel halves) instead of four: 36.9 KB per step instead of 73.7.
//   * the patch transform is three wave-items per step (rows 2 ap, 2 ap + 1 of B^T d B for all 32 tiles x a channel pair per
//     lane): 84 eight-byte patch reads per lane trio instead of 132, 8-byte V writes.
//   * the output transform A^T m A: its first half (over a, for the wave's own b) is lane-local; the six column sums of a
//     row cross LDS once (49 KB per output row) to the eight waves that finish a row exactly as conv_wf4_kernel does (same
//     operations in the same order: results are bit-identical to it).

constexpr int WF4X_THREADS = 768, WF4X_X_FLOATS = 12 * 4 * 64 * 4;

// rows 2 AP and 2 AP + 1 of B^T d B for a channel pair of one tile; V[b][kk][ap][tile][a & 1]
template <int AP, int rs, int ps>
__device__ __forceinline__ void wf4x_transform_pair(const float *P, float *V, int pbase, int vbase) {
    constexpr int A0 = 2 * AP, A1 = 2 * AP + 1;
    constexpr bool used[6][6] = {{1, 0, 1, 0, 1, 0}, {0, 1, 1, 1, 1, 0}, {0, 1, 1, 1, 1, 0},
                                 {0, 1, 1, 1, 1, 0}, {0, 1, 1, 1, 1, 0}, {0, 1, 0, 1, 0, 1}};
    // column by column, the next column's values requested before this one's are used (all thirty at once would not fit
    // beside the accumulators: 168 registers)
    wf4_v2 m0[6], m1[6], d[2][6];
    auto fetch = [&](int b, wf4_v2 (&dd)[6]) {
        const float *col = P + pbase + (b & 3) * ps + (b >> 2) * 4;
#pragma unroll
        for (int k = 0; k < 6; ++k)
            dd[k] = (used[A0][k] || used[A1][k]) ? *reinterpret_cast<const wf4_v2 *>(col + k * rs) : (wf4_v2){0.f, 0.f};
    };
#ifndef WF4X_DBUF
#define WF4X_DBUF 1
#endif
#if WF4X_DBUF
    fetch(0, d[0]);
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        if (b + 1 < 6) fetch(b + 1, d[(b + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        m0[b] = wf4_bt_row2<A0>(d[b & 1]);
        m1[b] = wf4_bt_row2<A1>(d[b & 1]);
        __builtin_amdgcn_sched_barrier(0);
    }
#else
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        fetch(b, d[0]);
        m0[b] = wf4_bt_row2<A0>(d[0]);
        m1[b] = wf4_bt_row2<A1>(d[0]);
        __builtin_amdgcn_sched_barrier(0);
    }
#endif
    wf4_v2 o0[6], o1[6];
    wf4_bt2(m0, o0);
    wf4_bt2(m1, o1);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        *reinterpret_cast<wf4_v2 *>(V + vbase + j * 768) = (wf4_v2){o0[j].x, o1[j].x};             // channel 2 cp
        *reinterpret_cast<wf4_v2 *>(V + vbase + j * 768 + 192) = (wf4_v2){o0[j].y, o1[j].y};       // channel 2 cp + 1
    }
}

// One row of B^T d B for a channel PAIR of one tile per lane (lanes = 32 tiles x 2 pairs, pair fastest: consecutive lanes read
// consecutive 8 bytes): packed arithmetic, a sixth of a chunk's transform in about forty vector instructions.
template <int A, int rs, int ps>
__device__ __forceinline__ void wf4x_transform_row2(const float *P, float *V, int pbase, int vbase) {
    constexpr bool used[6][6] = {{1, 0, 1, 0, 1, 0}, {0, 1, 1, 1, 1, 0}, {0, 1, 1, 1, 1, 0},
                                 {0, 1, 1, 1, 1, 0}, {0, 1, 1, 1, 1, 0}, {0, 1, 0, 1, 0, 1}};
    wf4_v2 d[6][6];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        const float *col = P + pbase + (b & 3) * ps + (b >> 2) * 4;
#pragma unroll
        for (int k = 0; k < 6; ++k)
            d[b][k] = used[A][k] ? *reinterpret_cast<const wf4_v2 *>(col + k * rs) : (wf4_v2){0.f, 0.f};
    }
    __builtin_amdgcn_sched_barrier(0);
    wf4_v2 m[6], o[6];
#pragma unroll
    for (int b = 0; b < 6; ++b) m[b] = wf4_bt_row2<A>(d[b]);
    wf4_bt2(m, o);
    float *v0 = V + vbase + (A >> 1) * 64 + (A & 1);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        v0[j * 768] = o[j].x;             // channel 2 cp
        v0[j * 768 + 192] = o[j].y;       // channel 2 cp + 1
    }
}

// One row of B^T d B for one channel of one tile per lane, in two halves: the patch values the row needs (3 or 4 per patch
// column) are read into registers early -- behind the previous step's MFMAs -- and turned into the six V values later, in
// the step's transform phase, by ~30 vector instructions that wait for nothing.
template <int A, int rs, int ps>
__device__ __forceinline__ void wf4x_row1_fetch(const float *P, int pbase, float (&dreg)[6][4]) {
    constexpr int kidx[6][4] = {{0, 2, 4, -1}, {1, 2, 3, 4}, {1, 2, 3, 4}, {1, 2, 3, 4}, {1, 2, 3, 4}, {1, 3, 5, -1}};
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        const float *col = P + pbase + (b & 3) * ps + (b >> 2) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (kidx[A][i] >= 0) dreg[b][i] = col[kidx[A][i] * rs];
    }
}
template <int A>
__device__ __forceinline__ void wf4x_row1_compute(const float (&dreg)[6][4], float *V, int vbase) {
    constexpr int kidx[6][4] = {{0, 2, 4, -1}, {1, 2, 3, 4}, {1, 2, 3, 4}, {1, 2, 3, 4}, {1, 2, 3, 4}, {1, 3, 5, -1}};
    float m[6], o[6];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        float d[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (kidx[A][i] >= 0) d[kidx[A][i]] = dreg[b][i];
        m[b] = wf4_bt_row<A>(d);
    }
    wf4_bt(m, o);
#pragma unroll
    for (int j = 0; j < 6; ++j) V[vbase + j * 768 + (A >> 1) * 64 + (A & 1)] = o[j];
}

// One row of B^T d B for one channel of one tile per lane (lanes = 16 tiles x 4 channels, channel fastest: four lanes read one
// 16-byte patch cell, a wave 256 contiguous bytes): a twelfth of a chunk's transform -- about thirty vector instructions.
template <int A, int rs, int ps>
__device__ __forceinline__ void wf4x_transform_row1(const float *P, float *V, int pbase, int vbase) {
    constexpr bool used[6][6] = {{1, 0, 1, 0, 1, 0}, {0, 1, 1, 1, 1, 0}, {0, 1, 1, 1, 1, 0},
                                 {0, 1, 1, 1, 1, 0}, {0, 1, 1, 1, 1, 0}, {0, 1, 0, 1, 0, 1}};
    float d[6][6];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        const float *col = P + pbase + (b & 3) * ps + (b >> 2) * 4;
#pragma unroll
        for (int k = 0; k < 6; ++k) d[b][k] = used[A][k] ? col[k * rs] : 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);
    float m[6], o[6];
#pragma unroll
    for (int b = 0; b < 6; ++b) m[b] = wf4_bt_row<A>(d[b]);
    wf4_bt(m, o);
#pragma unroll
    for (int j = 0; j < 6; ++j) V[vbase + j * 768 + (A >> 1) * 64 + (A & 1)] = o[j];
}

// The same two rows for ONE channel of one tile per lane (lanes = 16 tiles x 4 channels, channel fastest: four lanes read one
// 16-byte patch cell, a wave 256 contiguous bytes -- conflict-free 4-byte reads): a third of the packed form's registers, so
// that a phase's fifteen patch values can all be in flight beside the accumulators.  Same formulas, same roundings.
template <int AP, int rs, int ps>
__device__ __forceinline__ void wf4x_transform_pair1(const float *P, float *V, int pbase, int vbase) {
    constexpr int A0 = 2 * AP, A1 = 2 * AP + 1;
    constexpr bool used[6][6] = {{1, 0, 1, 0, 1, 0}, {0, 1, 1, 1, 1, 0}, {0, 1, 1, 1, 1, 0},
                                 {0, 1, 1, 1, 1, 0}, {0, 1, 1, 1, 1, 0}, {0, 1, 0, 1, 0, 1}};
    float m0[6], m1[6];
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
        float d[3][6];
#pragma unroll
        for (int bb = 0; bb < 3; ++bb) {
            const int b = 3 * ph + bb;
            const float *col = P + pbase + (b & 3) * ps + (b >> 2) * 4;
#pragma unroll
            for (int k = 0; k < 6; ++k) d[bb][k] = (used[A0][k] || used[A1][k]) ? col[k * rs] : 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int bb = 0; bb < 3; ++bb) {
            m0[3 * ph + bb] = wf4_bt_row<A0>(d[bb]);
            m1[3 * ph + bb] = wf4_bt_row<A1>(d[bb]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float o0[6], o1[6];
    wf4_bt(m0, o0);
    wf4_bt(m1, o1);
#pragma unroll
    for (int j = 0; j < 6; ++j) *reinterpret_cast<float2 *>(V + vbase + j * 768) = make_float2(o0[j], o1[j]);
}

template <int LBC>
__device__ __forceinline__ void conv_wf4x_body(const Wf4Args &p) {
    constexpr int S = (1 << LBC) + 1;               // 16-byte cells per x phase: compile time, so every patch read is base + immediate
    // separate LDS objects: the compiler orders LDS-DMA against later LDS accesses object by object
    __shared__ __attribute__((aligned(16))) float Vs0[WF4_V_FLOATS], Vs1[WF4_V_FLOATS];      // [6 b][4 kk][3 ap][32 tiles][2]
    __shared__ __attribute__((aligned(16))) float Ps0[WF4_P_FLOATS], Ps1[WF4_P_FLOATS], Ps2[WF4_P_FLOATS], Ps3[WF4_P_FLOATS];      // [cells][4]
    __shared__ __attribute__((aligned(16))) float Xs[WF4X_X_FLOATS];                         // [12 waves][4 blocks][64 lanes][4]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wh = wave >= 6 ? 1 : 0, wb = wave - 6 * wh;      // channel half, frequency column
    const int li = lane & 15, lk = lane >> 4;

    // ---- which block: cout block fastest (blocks that share input pixels are neighbours), then columns, rows, images ----
    unsigned t1, coutblk, t2, colblk, ngrp, rowblk;
    p.divCoB.divmod(blockIdx.x, t1, coutblk);
    p.divCb.divmod(t1, t2, colblk);
    p.divRb.divmod(t2, ngrp, rowblk);
    const int BRm = (1 << p.lBR) - 1, BCm = (1 << LBC) - 1, lT = p.lBR + LBC;
    const int n0 = (int)ngrp << (5 - lT), ty0 = (int)rowblk << p.lBR, tx0 = (int)colblk << LBC;
    const int HW = p.H * p.W;

    const __amdgpu_buffer_rsrc_t ursrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.u), 0, p.u_bytes, 0x00020000);
    constexpr int OOB = (int)0x80000000;
    typedef __attribute__((address_space(3))) float lds_float;

    // ---- patch cells this thread fetches every chunk: cell tid, and cell 768 + tid for the first four waves ----
    int pvoff[2];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const unsigned ci = (unsigned)(ps * WF4X_THREADS + tid);
        pvoff[ps] = OOB;
        if (ci < (unsigned)p.cells) {
            unsigned nb, rem, r_, rem2, m, s;
            p.divPlane.divmod(ci, nb, rem);
            p.div4S.divmod(rem, r_, rem2);
            p.divS.divmod(rem2, m, s);
            const int n = n0 + (int)nb, h = 4 * ty0 + (int)r_ - 1, w = 4 * tx0 + (int)(4 * s + m) - 1;
            if (n < p.N && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W)
                pvoff[ps] = (int)(((unsigned)(n * p.Cq) * (unsigned)HW + (unsigned)(h * p.W + w)) << 4);
        }
    }
    // every wave issues the first pass (lanes past the last cell carry the out-of-range offset and zero-fill cells nobody
    // reads), waves 0-3 the second where the patch has more than 768 cells: the request count of a step does not depend on
    // run-time conditions the compiler would have to assume false when it places its waits
    const bool pass1 = wave < 4 && WF4X_THREADS + wave * 64 < p.cells;
    // The LDS-DMA goes out as inline assembly: the compiler orders an LDS-DMA builtin against every later access to an LDS
    // object it may write -- across the loop's back edge that is s_waitcnt vmcnt(0) in front of the transform's patch reads,
    // i.e. no request could stay in flight over a step.  Ordering is this kernel's business (wait_barrier below).
    typedef int wf4x_i4 __attribute__((ext_vector_type(4)));
    const wf4x_i4 xres = {(int)(unsigned)(uintptr_t)p.x, (int)((unsigned)((uintptr_t)p.x >> 32) & 0xffffu), (int)p.x_bytes, 0x00020000};
    auto load_p_piece = [&](int c, int buf, int ps) {
        const int soff = (c * HW) << 4;
        float *Pb = buf == 0 ? Ps0 : buf == 1 ? Ps1 : buf == 2 ? Ps2 : Ps3;
        if constexpr (WF4_KNOCK & 2) return;
        // (the wave's destination as a scalar: buffer base + 1 KB per wave-slot)
        const unsigned dst = (unsigned)(uintptr_t)(lds_float *)Pb + (unsigned)(ps * WF4X_THREADS + wave * 64) * 16u;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                     :: "s"(dst), "v"(pvoff[ps]), "s"(xres), "s"(soff) : "memory");
    };
    // filter fragment: 12 floats [cb][a] of (channel 32 wh + 16 cb + li, k = lk), the wave's 3 KB contiguous
    const int a_voff = lane * 48;
    auto a_soff = [&](int c) {
        return (int)((((unsigned)coutblk * (unsigned)p.nchunks + (unsigned)c) * (unsigned)WF4_A_FLOATS + (unsigned)wave * 768u) * 4u);
    };
    auto load_a = [&](int c, int q) -> float4 {
        if constexpr (WF4_KNOCK & 1) return make_float4(1.f, 2.f, 3.f, 4.f);
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ursrc, a_voff + 16 * q, a_soff(c), 0));
    };

    // ---- transform: twelve pieces, one per wave: row (wave % 6) of B^T d B for one half of the block's tiles, lane = (tile,
    //      channel), in a phase of its own in front of the step's MFMAs, on values fetched a step earlier.  Why a phase: fp32
    //      MFMAs and vector-ALU instructions do not overlap on a SIMD (tools/ubench/mfma_valu.hip: a v_fma from the wave that
    //      issues the MFMAs adds ~3 cycles to the stream, one from ANOTHER wave of the SIMD ~20; tools/wf4_trace.py: a transform
    //      item of 130 instructions beside two MFMA waves took 2200-2600 cycles whatever its priority), so the cheapest place
    //      for the transform's arithmetic is where no MFMA is in flight -- and its LDS reads (66 KB a step, 500+ cycles of the
    //      LDS array) belong behind the MFMAs, which need no LDS ----
    const int trow = wave >= 6 ? wave - 6 : wave;
    const int ttile = (wave >= 6 ? 16 : 0) + (lane >> 2), tch = lane & 3;
    constexpr int rs = 16 * S, psz = 4 * S;
    int pb2, vb2;
    {
        const int t_nb = ttile >> lT, t_r = (ttile >> LBC) & BRm, t_c = ttile & BCm;
        pb2 = tch + (((t_nb * p.R + 4 * t_r) * 4) * S + t_c) * 4;
        vb2 = tch * 192 + ttile * 2;
    }
    float dreg[6][4];
    auto fetch = [&](int pbuf) {
        if constexpr (WF4_KNOCK & 4) return;
        const float *Pb = pbuf == 0 ? Ps0 : pbuf == 1 ? Ps1 : pbuf == 2 ? Ps2 : Ps3;
        switch (trow) {
        case 0: wf4x_row1_fetch<0, rs, psz>(Pb, pb2, dreg); break;
        case 1: wf4x_row1_fetch<1, rs, psz>(Pb, pb2, dreg); break;
        case 2: wf4x_row1_fetch<2, rs, psz>(Pb, pb2, dreg); break;
        case 3: wf4x_row1_fetch<3, rs, psz>(Pb, pb2, dreg); break;
        case 4: wf4x_row1_fetch<4, rs, psz>(Pb, pb2, dreg); break;
        default: wf4x_row1_fetch<5, rs, psz>(Pb, pb2, dreg); break;
        }
    };
    auto transform = [&](int vbuf) {
        if constexpr (WF4_KNOCK & 4) return;
        float *Vb = vbuf ? Vs1 : Vs0;
        switch (trow) {
        case 0: wf4x_row1_compute<0>(dreg, Vb, vb2); break;
        case 1: wf4x_row1_compute<1>(dreg, Vb, vb2); break;
        case 2: wf4x_row1_compute<2>(dreg, Vb, vb2); break;
        case 3: wf4x_row1_compute<3>(dreg, Vb, vb2); break;
        case 4: wf4x_row1_compute<4>(dreg, Vb, vb2); break;
        default: wf4x_row1_compute<5>(dreg, Vb, vb2); break;
        }
    };

    f32x4 acc[2][2][6];      // [cb][tb][a]
#pragma unroll
    for (int i = 0; i < 24; ++i) (&acc[0][0][0])[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int b_off = (wb * 4 + lk) * 192 + li * 2;      // V[b = wb][kk = lk][ap][tile li (+ 16 tb)][0]

    // The filter fragment (cb0 a0-3 | cb0 a4-5, cb1 a0-1 | cb1 a2-5) rolls: a quad is requested again, for the next chunk, right
    // behind the eight MFMAs that were its last readers, and is first read two thirds of a step later -- one register set, no copy.
    float4 af[3];
    auto mma = [&](int buf, int cnext, int pfetch) {
        const float *Vp = (buf ? Vs1 : Vs0) + b_off;
        wf4_v2 v[3][2];
        if constexpr (WF4_KNOCK & 16) {
#pragma unroll
            for (int ap = 0; ap < 3; ++ap) v[ap][0] = v[ap][1] = (wf4_v2){1.f, (float)lane};
        } else {
#pragma unroll
            for (int ap = 0; ap < 3; ++ap) {
                v[ap][0] = *reinterpret_cast<const wf4_v2 *>(Vp + ap * 64);
                v[ap][1] = *reinterpret_cast<const wf4_v2 *>(Vp + ap * 64 + 32);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        fetch(pfetch);                                  // the next transform's patch values: they land behind the MFMAs
        __builtin_amdgcn_sched_barrier(0);
        auto mm = [&](int cb, int a, float av) {
            const float b0 = (a & 1) ? v[a >> 1][0].y : v[a >> 1][0].x, b1 = (a & 1) ? v[a >> 1][1].y : v[a >> 1][1].x;
            if constexpr (WF4_KNOCK & 8) {
                acc[cb][0][a].x += av * b0; acc[cb][1][a].x += av * b1;
            } else {
                acc[cb][0][a] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0, acc[cb][0][a], 0, 0, 0);
                acc[cb][1][a] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1, acc[cb][1][a], 0, 0, 0);
            }
        };
        mm(0, 0, af[0].x); mm(0, 1, af[0].y); mm(0, 2, af[0].z); mm(0, 3, af[0].w);
        __builtin_amdgcn_sched_barrier(0);
        af[0] = load_a(cnext, 0);
        __builtin_amdgcn_sched_barrier(0);
        mm(0, 4, af[1].x); mm(0, 5, af[1].y); mm(1, 0, af[1].z); mm(1, 1, af[1].w);
        __builtin_amdgcn_sched_barrier(0);
        af[1] = load_a(cnext, 1);
        __builtin_amdgcn_sched_barrier(0);
        mm(1, 2, af[2].x); mm(1, 3, af[2].y); mm(1, 4, af[2].z); mm(1, 5, af[2].w);
        __builtin_amdgcn_sched_barrier(0);
        af[2] = load_a(cnext, 2);
    };

    // Barrier that leaves the k most recent vector-memory requests of this wave in flight (__syncthreads() waits for all of
    // them: the requests of a step would have to land within the step, and a step is shorter than their latency under load).
    auto wait_barrier = [&](int k) {
        switch (k) {
        case 0: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
        }
    };
    // LDS-DMA instructions this wave issues per patch
    const int nd = pass1 ? 2 : 1;
    auto load_patch = [&](int c, int buf) {
        load_p_piece(c, buf, 0);
        if (pass1) load_p_piece(c, buf, 1);
    };

    // ---- prologue: patches 0-3 requested, filter fragment 0 requested, chunk 0 transformed, chunk 1's patch values fetched ----
    load_patch(0, 0);
    load_patch(min(1, p.nchunks - 1), 1);
    load_patch(min(2, p.nchunks - 1), 2);
    load_patch(min(3, p.nchunks - 1), 3);
    af[0] = load_a(0, 0); af[1] = load_a(0, 1); af[2] = load_a(0, 2);
    wait_barrier(0);
    fetch(0);
    transform(0);
    fetch(1);
    wait_barrier(0);
    // One K step (chunk c; V parity CV = c % 2, patch slot CP = c % 4).  Transform phase: every wave turns the patch values it
    // holds (chunk c + 1) into its piece of V[CV ^ 1]; barrier.  MFMA phase: the wave reads its V fragment, fetches its patch
    // values of chunk c + 2 (requested three steps ago), requests its share of chunk c + 4's patch (into the buffer whose values
    // were fetched two steps ago) and multiplies chunk c, re-requesting each filter quad for chunk c + 1 behind its last MFMA.
    // It leaves the step with only the requests it has just made in flight.
#ifndef WF4X_PRIO
#define WF4X_PRIO 0
#endif
#ifdef WF4X_TRACE
    unsigned ts8_0 = 0, ts8_1 = 0, ts8_2 = 0, ts8_3 = 0, ts8_4 = 0, ts9_0 = 0, ts9_1 = 0, ts9_2 = 0, ts9_3 = 0, ts9_4 = 0;
#endif
    auto kstep = [&](auto cv_, auto cp_, int c) {
        constexpr int CV = decltype(cv_)::value, CP = decltype(cp_)::value;
        const bool more = c + 1 < p.nchunks;
        // (requested also past the last chunk -- the last chunk again, into registers / a buffer nobody reads any more: a
        //  conditional request would make the compiler wait for the newest requests wherever it waits for an older one)
#ifdef WF4X_TRACE
#define WF4X_STAMP(k) do { if (blockIdx.x == WF4X_TRACE) { if (c == 8) ts8_##k = (unsigned)__builtin_readcyclecounter(); \
                                                            else if (c == 9) ts9_##k = (unsigned)__builtin_readcyclecounter(); } } while (0)
        WF4X_STAMP(0);
#endif
        if (more) transform(CV ^ 1);
        asm volatile("s_barrier" ::: "memory");      // no MFMA of this step before the SIMD's transform pieces are done
#ifdef WF4X_TRACE
        WF4X_STAMP(1);
#endif
        load_patch(min(c + 4, p.nchunks - 1), CP);
        mma(CV, min(c + 1, p.nchunks - 1), (CP + 2) & 3);
#ifdef WF4X_TRACE
        WF4X_STAMP(2);
#endif
        const int issued = 3 + nd;
#ifdef WF4X_TRACE
        WF4X_STAMP(3);
#endif
        wait_barrier(issued);
#ifdef WF4X_TRACE
        WF4X_STAMP(4);
#endif
    };
    if (WF4X_PRIO && (wave & 4)) __builtin_amdgcn_s_setprio(WF4X_PRIO);
    {
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        using I3 = std::integral_constant<int, 3>;
        for (int c = 0; c < p.nchunks; c += 4) {
            kstep(I0{}, I0{}, c);
            if (c + 1 < p.nchunks) kstep(I1{}, I1{}, c + 1);
            if (c + 2 < p.nchunks) kstep(I0{}, I2{}, c + 2);
            if (c + 3 < p.nchunks) kstep(I1{}, I3{}, c + 3);
        }
    }
    if (WF4X_PRIO) __builtin_amdgcn_s_setprio(0);
    wait_barrier(0);          // nothing of this workgroup's may still be on its way into LDS
#ifdef WF4X_TRACE
    if (blockIdx.x == WF4X_TRACE && lane == 0) {
        // probe build: the stamps of steps 8 and 9 (low 32 bits) over the last floats of the INPUT
        unsigned *dbg = reinterpret_cast<unsigned *>(const_cast<float *>(p.x)) + (p.x_bytes / 4 - 256) + wave * 16;
        dbg[0] = ts8_0; dbg[1] = ts8_1; dbg[2] = ts8_2; dbg[3] = ts8_3; dbg[4] = ts8_4;
        dbg[5] = ts9_0; dbg[6] = ts9_1; dbg[7] = ts9_2; dbg[8] = ts9_3; dbg[9] = ts9_4;
    }
#endif

    // ---- output transform: first half lane-local, the six column sums of a row through Xs to the finishing waves ----
    const bool fin = wb < 4;                          // eight finishing waves: (wh, block cb = wb / 2, tb = wb % 2)
    const int fcb = wb >> 1, ftb = wb & 1;
    const int wm = wh * 2 + fcb;                      // 16-channel block of the 64
    const int coq = (int)coutblk * 16 + wm * 4 + lk;
    const int cqc = min(coq, p.Coq - 1);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.ep.res), 0, p.ep.res ? p.y_bytes : 0u, 0x00020000);
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f), one4 = make_float4(1.f, 1.f, 1.f, 1.f);
    const bool plain = !p.ep.bias && p.ep.scale && p.ep.shift && p.ep.act == 1 && !(p.ep.res && p.ep.res_post);
    const float4 scale = plain ? reinterpret_cast<const float4 *>(p.ep.scale)[cqc] : one4;
    const float4 shift = plain ? reinterpret_cast<const float4 *>(p.ep.shift)[cqc] : z4;
    // the K loop is over: V0 / V1 are free -- four exchange buffers of 4.25 KB in each
    const int fw = wh * 4 + wb;
    float4 *xb = reinterpret_cast<float4 *>(fw < 4 ? Vs0 : Vs1) + (fw & 3) * (4 * 68);
    const int te = lane >> 2, be = lane & 3;
    const int oj2 = ftb * 16 + te;
    const int n2 = n0 + (oj2 >> lT), ty2 = ty0 + ((oj2 >> LBC) & BRm), tx2 = tx0 + (oj2 & BCm);
    const int x2 = tx2 * 4 + be, cq0 = (int)coutblk * 16 + wm * 4;
    const bool ok2 = n2 < p.N && ty2 < p.th && tx2 < p.tw && x2 < p.W;
    float4 *Xw = reinterpret_cast<float4 *>(Xs) + wave * 256 + lane;                      // this wave's four blocks
    const float4 *Xr = reinterpret_cast<const float4 *>(Xs) + (wh * 6) * 256 + (fcb * 2 + ftb) * 64 + lane;
    // the four rows' sums of every block first (16 quads): the 96 accumulator registers are dead before the tail needs its own
    float4 sums[4][4];      // [row A][block]
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            wf4_v2 mlo[6], mhi[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                mlo[a] = __builtin_shufflevector(acc[cb][tb][a], acc[cb][tb][a], 0, 1);
                mhi[a] = __builtin_shufflevector(acc[cb][tb][a], acc[cb][tb][a], 2, 3);
            }
            const wf4_v2 l0 = wf4_at_row2<0>(mlo), l1 = wf4_at_row2<1>(mlo), l2 = wf4_at_row2<2>(mlo), l3 = wf4_at_row2<3>(mlo);
            const wf4_v2 h0 = wf4_at_row2<0>(mhi), h1 = wf4_at_row2<1>(mhi), h2 = wf4_at_row2<2>(mhi), h3 = wf4_at_row2<3>(mhi);
            sums[0][cb * 2 + tb] = make_float4(l0.x, l0.y, h0.x, h0.y);
            sums[1][cb * 2 + tb] = make_float4(l1.x, l1.y, h1.x, h1.y);
            sums[2][cb * 2 + tb] = make_float4(l2.x, l2.y, h2.x, h2.y);
            sums[3][cb * 2 + tb] = make_float4(l3.x, l3.y, h3.x, h3.y);
        }
    auto row = [&](auto first, auto res, auto pl) {
        constexpr int A = decltype(first)::value;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) Xw[blk * 64] = sums[A][blk];
        __syncthreads();
        if (fin) {
            const int yy = ty2 * 4 + A;
            int off[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                off[q] = (ok2 && yy < p.H && cq0 + q < p.Coq)
                             ? (int)((((unsigned)(n2 * p.Coq + cq0 + q) * (unsigned)p.H + (unsigned)yy) * (unsigned)p.W + (unsigned)x2) << 4) : OOB;
            wf4_v2 slo[6], shi[6];
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                const float4 s = Xr[b * 256];
                slo[b] = (wf4_v2){s.x, s.y};
                shi[b] = (wf4_v2){s.z, s.w};
            }
            wf4_row_finish<decltype(res)::value, decltype(pl)::value>(p, slo, shi, cqc, scale, shift, xb, lk * 16 + li, be * 68 + te,
                                                                     yrsrc, rrsrc, off);
        }
        if (A < 3) __syncthreads();                   // the next row's sums overwrite Xs
    };
    auto rows = [&](auto res, auto pl) {
        row(std::integral_constant<int, 0>{}, res, pl);
        row(std::integral_constant<int, 1>{}, res, pl);
        row(std::integral_constant<int, 2>{}, res, pl);
        row(std::integral_constant<int, 3>{}, res, pl);
    };
    if (plain) {
        if (p.ep.res) rows(std::true_type{}, std::true_type{});
        else rows(std::false_type{}, std::true_type{});
    } else {
        if (p.ep.res) rows(std::true_type{}, std::false_type{});
        else rows(std::false_type{}, std::false_type{});
    }
}

template <int LBC>
__global__ void __launch_bounds__(WF4X_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) conv_wf4x_kernel(const Wf4Args p) {
    conv_wf4x_body<LBC>(p);
}

// filter: OIHW 3x3 -> u[cout block][chunk][wave = 6 h + b][kk][i][cb][a] = (G g G^T)[a][b] of channel (64 blk + 32 h + 16 cb + i,
// 4 chunk + kk); zero beyond Cout
__global__ void __launch_bounds__(256) wf4x_filter_kernel(const float *w, float *u, unsigned total, int Cin, int Cout) {
    const unsigned i = blockIdx.x * 256 + threadIdx.x;   // (co, c) pair
    if (i >= total) return;
    const int co = (int)(i / (unsigned)Cin), c = (int)(i - (unsigned)co * Cin);
    const float *g = w + (size_t)i * 9;
    float t[6][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float g0 = g[j], g1 = g[3 + j], g2 = g[6 + j];
        t[0][j] = g0 * 0.25f;
        t[1][j] = -(g0 + g1 + g2) * (1.f / 6.f);
        t[2][j] = (-g0 + g1 - g2) * (1.f / 6.f);
        t[3][j] = g0 * (1.f / 24.f) + g1 * (1.f / 12.f) + g2 * (1.f / 6.f);
        t[4][j] = g0 * (1.f / 24.f) - g1 * (1.f / 12.f) + g2 * (1.f / 6.f);
        t[5][j] = g2;
    }
    const int nchunks = Cin / 4;
    float *up = u + ((size_t)(co >> 6) * nchunks + (c >> 2)) * WF4_A_FLOATS + ((co >> 5) & 1) * (6 * 768) +
                ((c & 3) * 16 + (co & 15)) * 12 + ((co >> 4) & 1) * 6;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        const float g0 = t[a][0], g1 = t[a][1], g2 = t[a][2];
        up[0 * 768 + a] = g0 * 0.25f;
        up[1 * 768 + a] = -(g0 + g1 + g2) * (1.f / 6.f);
        up[2 * 768 + a] = (-g0 + g1 - g2) * (1.f / 6.f);
        up[3 * 768 + a] = g0 * (1.f / 24.f) + g1 * (1.f / 12.f) + g2 * (1.f / 6.f);
        up[4 * 768 + a] = g0 * (1.f / 24.f) - g1 * (1.f / 12.f) + g2 * (1.f / 6.f);
        up[5 * 768 + a] = g2;
    }
}
