// np_eventalign_kernel.hip -- the eventalign segment chain on the device: align_read_to_ref
// (src/alignment/nanopolish_eventalign.cpp:612-826) for a batch of reads, one wavefront per read.
//
// The reference realigns a read to its reference in ~100-base segments.  Every segment is one profile_hmm_align (Viterbi
// fill + back-track, src/hmm/nanopolish_profile_hmm_r9.cpp:73-204, r9.inl:130-197, flags 0) over the events between the
// segment's start event and the closest event of its last aligned base; of the aligned states only the first ~50 are
// emitted, and the next segment starts at the last emitted (event, reference k-mer).  The chain is data-dependent from
// segment to segment but independent between reads, so a persistent wave walks one read's chain from end to end:
//   * segment geometry from the read's CIGAR without materialising aligned pairs (np_cigar.h; get_end_pair :196-205 is a
//     "last aligned pair with ref_pos <= x" search), closest events from the read's event map (np_device.h);
//   * Viterbi fill as an anti-diagonal sweep: lane j owns k-mer blocks 2j, 2j+1 (a segment has <= 96 k-mers), computes row
//     t - j at step t; previous row in registers, left neighbour through DPP; six candidates in HMMMovementType order,
//     later index wins ties; only back-pointers leave the wave: 6 bits per block and row (M: 3, B: 1, K: 2), one byte,
//     one coalesced 128-byte line per sweep STEP (cell (row r, k-mer b) sits in line r + b / 2 at byte b), into a per-wave
//     scratch that stays in L2; events reach the lanes by one block load per 64 steps + v_readlane + a DPP shift;
//   * back-track with a wave-uniform (scalar) state over 32 lines at a time staged in LDS, every visited state appended
//     to a per-wave path list (64 entries per coalesced flush); then the wave reads the tail of the list back and emits
//     it 64 entries at a time (ballot + prefix count reproduces the reference's "first 50 that are not K and not the
//     start event" cut).
// Output rows are (ref_position relative to the record's pos, event_idx, state 'M'/'B'); ref_kmer / model_kmer of the TSV
// follow from them on the host (nanopolish_amd/eventalign.py).
#include "np_kernels.h"
#include "np_cigar.h"

#define NP_EA_ROW_BYTES 128          // back-pointer bytes per sweep step (lane l's two blocks at bytes 2l, 2l + 1)
#define NP_EA_MAX_KMERS 128
#define NP_EA_CHUNK 32               // back-pointer lines staged in LDS per back-track pass (4 KB per wave)

namespace {

__device__ __forceinline__ float readlane_f32(float v, int l)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

__device__ __forceinline__ uint32_t base_code(char c) { return c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 0u; }   // disambiguated to ACGT upstream

// The Viterbi sweep of one segment, as a separate (not inlined) function: the chain loop around it keeps ~100 wave-uniform values
// alive (CIGAR view, read record, output cursors); inlined, the register allocator spills some of them INSIDE this loop.  Called
// once per segment, the caller's state is parked around the call instead and the sweep gets the registers to itself.
struct ea_trans { float mm_self, mb, mk, mm_next, bb, bk, bm_next, bm_self, kk, km; };
__device__ __attribute__((noinline)) float ea_fill(const np_gauss g0, const np_gauss g1, const ea_trans tr, const float flank0_in,
                                                   const float* __restrict__ ev, uint8_t* __restrict__ bp, const int e_start, const int stride,
                                                   const int e, const int n, const int lane)
{
    const np_gauss g[2] = {g0, g1};
    const float lp_mm_self = tr.mm_self, lp_mb = tr.mb, lp_mk = tr.mk, lp_mm_next = tr.mm_next, lp_bb = tr.bb, lp_bk = tr.bk,
                lp_bm_next = tr.bm_next, lp_bm_self = tr.bm_self, lp_kk = tr.kk, lp_km = tr.km;
    const int lanes_used = (n + 1) >> 1;
    float M0 = NP_NEG_INF, M1 = NP_NEG_INF, B0 = NP_NEG_INF, B1 = NP_NEG_INF, K0 = NP_NEG_INF, K1 = NP_NEG_INF;   // row r-1 of this lane's two blocks
    float oM = NP_NEG_INF, oB = NP_NEG_INF, oK = NP_NEG_INF;                                                      // row r-1 of the block to the left
    const int steps = e + lanes_used - 1;
    const float flank0 = flank0_in;
    const int end_lane = (n - 1) >> 1, end_c = (n - 1) & 1;
    // The sweep is branch-free: every lane updates its two blocks at every step.  A lane that is before its first row
    // only moves -inf around (row 0 is all -inf), one that is past its last row or owns no block computes values nobody
    // reads.  Loads and stores go through range-checked descriptors sized to the segment: an event outside it reads
    // as 0 (it only feeds such cells), a back-pointer row outside [0, e) is dropped by the hardware -- so the
    // per-lane addresses are plain running offsets (one add each per step) with no clamp, compare or select.
    const __amdgpu_buffer_rsrc_t evr = make_rsrc(ev + (stride > 0 ? e_start : e_start - (e - 1)), (uint32_t)e * 4u);
    // Events: every lane walks the same sequence of events, lane j one step behind lane j - 1.  So only lane 0 needs a new
    // event per step and the others take their left neighbour's previous one (one DPP shift): the wave fetches 64 events
    // at a time with one coalesced load (lane i holds row 64 * blk + i + 1), a block ahead, and each step reads its event
    // out of that register with v_readlane -- no memory latency inside the sweep.
    auto ev_off = [&](int idx) { return stride > 0 ? 4 * idx : 4 * (e - 1 - idx); };     // byte offset of 0-based row idx
    float ecur = buf_f32(evr, ev_off(lane)), enxt = buf_f32(evr, ev_off(lane + 64));
    float x = 0.0f;
    for (int t = 1; t <= steps; ++t) {
        const float nM = np_wave_shr1(M1, NP_NEG_INF), nB = np_wave_shr1(B1, NP_NEG_INF), nK = np_wave_shr1(K1, NP_NEG_INF);   // lane 0: block -1 = -inf
        const int ti = (t - 1) & 63;
        if (ti == 0 && t > 1) { ecur = enxt; enxt = buf_f32(evr, ev_off(t - 1 + 64 + lane)); }
        const float x0 = readlane_f32(ecur, ti);                // the event of row t: lane 0's at this step
        x = np_wave_shr1(x, x0);
        // HMT_FROM_SOFT (flags 0: first event only, r9.inl:361-363) reaches block 0 of row 1: lane 0 at step 1
        const float soft_t = t == 1 ? flank0 : NP_NEG_INF;      // (scalar)
        const float soft = lane == 0 ? soft_t : NP_NEG_INF;
        uint32_t packed;
        {
            // ---- block 2*lane: left neighbour = previous lane's block (nM.. row r, oM.. row r-1) ----
            const float em = np_emission(x, g[0]);
            const float a0 = lp_mm_self + M0, a1 = lp_mm_next + oM, a2 = lp_bm_self + B0, a3 = lp_bm_next + oB, a4 = lp_km + oK;
            const float v = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(a0, a1), __builtin_fmaxf(a2, a3)), __builtin_fmaxf(a4, soft));
            uint32_t from = (a1 == v) ? 1u : 0u;               // the largest index whose candidate equals the maximum
            from = (a2 == v) ? 2u : from; from = (a3 == v) ? 3u : from; from = (a4 == v) ? 4u : from; from = (soft == v) ? 5u : from;
            const float newM = v + em;
            // (the B and K states emit 0: the reference's `+ lp_emission` leaves every value it can meet here unchanged)
            const float b0 = lp_mb + M0, b2 = lp_bb + B0;
            const float newB = __builtin_fmaxf(b0, b2);
            const uint32_t bbit = (b2 >= b0) ? 8u : 0u;
            const float k1 = lp_mk + nM, k3 = lp_bk + nB, k4 = lp_kk + nK;
            const float newK = __builtin_fmaxf(__builtin_fmaxf(k1, k3), k4);
            uint32_t kbits = (k3 == newK) ? 16u : 0u; kbits = (k4 == newK) ? 32u : kbits;
            packed = from | bbit | kbits;
            // ---- block 2*lane + 1: left neighbour = the block just computed (row r) and its previous row ----
            const float em1 = np_emission(x, g[1]);
            const float c0 = lp_mm_self + M1, c1 = lp_mm_next + M0, c2 = lp_bm_self + B1, c3 = lp_bm_next + B0, c4 = lp_km + K0;
            const float w = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(c0, c1), __builtin_fmaxf(c2, c3)), c4);
            uint32_t from1 = (c1 == w) ? 1u : 0u;
            from1 = (c2 == w) ? 2u : from1; from1 = (c3 == w) ? 3u : from1; from1 = (c4 == w) ? 4u : from1;
            const float newM1 = w + em1;
            const float d0 = lp_mb + M1, d2 = lp_bb + B1;
            const float newB1 = __builtin_fmaxf(d0, d2);
            const uint32_t bbit1 = (d2 >= d0) ? 8u : 0u;
            const float j1 = lp_mk + newM, j3 = lp_bk + newB, j4 = lp_kk + newK;
            const float newK1 = __builtin_fmaxf(__builtin_fmaxf(j1, j3), j4);
            uint32_t kbits1 = (j3 == newK1) ? 16u : 0u; kbits1 = (j4 == newK1) ? 32u : kbits1;
            packed |= (from1 | bbit1 | kbits1) << 8;
            M0 = newM; B0 = newB; K0 = newK; M1 = newM1; B1 = newB1; K1 = newK1;
        }
        oM = nM; oB = nB; oK = nK;
        // back-pointers are laid out by sweep STEP, not by lattice row: step t writes one contiguous 128-byte line (lane l's two
        // blocks at bytes 2l, 2l + 1), whatever row each lane is on.  Cell (row r, k-mer b) therefore lives in line r + b / 2 at
        // byte b; lines and bytes that belong to no cell (lanes before their first / past their last row, or without a block)
        // hold values nobody reads.  (Row-major, the same store touched 64 different cache lines per step.)
        *(uint16_t*)(bp + (size_t)(t - 1) * NP_EA_ROW_BYTES + 2 * lane) = (uint16_t)packed;
    }
    // the lane that owns the last k-mer computes its last row in the last step, so its registers still hold it
    const float end_m = end_c ? M1 : M0;
    return __shfl(end_m, end_lane, 64);
}

__global__ void __launch_bounds__(64, 5) np_eventalign_chain_kernel(np_ea_args a)
{
    __shared__ uint4 stage[NP_EA_CHUNK * NP_EA_ROW_BYTES / 16];
    const int lane = threadIdx.x;
    const int wave_slot = blockIdx.x;
    uint8_t* __restrict__ bp = a.bp + (size_t)wave_slot * a.bp_stride;
    uint32_t* __restrict__ path = a.path + (size_t)wave_slot * a.path_stride;
    const int k = a.k;

    for (;;) {
        const int ri = __builtin_amdgcn_readfirstlane((int)atomicAdd(a.counter, lane == 0 ? 1u : 0u));
        if (ri >= a.n_reads) break;
        const np_read_dev* rd = a.reads + ri;
        const float* __restrict__ ev = a.event_mean + rd->event_off;
        const int32_t* __restrict__ ms = a.map_start + rd->rank_off;
        const int K = (int)rd->n_kmers;
        const char* __restrict__ ref = a.genome + a.ref_begin[ri];
        const int ref_n = a.ref_len[ri];
        const int rl = a.read_len[ri];
        const bool rc = a.read_rc[ri] != 0;
        const cig_view cv{a.cigar + a.cigar_off[ri], a.op_ref + a.cigar_off[ri] + ri, a.op_read + a.cigar_off[ri] + ri,
                          (int)(a.cigar_off[ri + 1] - a.cigar_off[ri])};
        const int64_t o0 = a.out_off[ri];
        const int out_cap = (int)(a.out_off[ri + 1] - o0);
        int n_out = 0, n_calls = 0, status = NP_EA_OK;
        unsigned long long cells = 0ull, rows = 0ull, kmers = 0ull;          // lattice cells / rows of the read's segments (statistics for the roofline)

        // aligned pairs trimmed to read_pos <= max_kmer_idx (trim_aligned_pairs_to_kmer, :167-177)
        const int max_kmer_idx = rl - k;
        int q_first = 0, r_first = 0, q_last = 0, r_last = 0;
        // a read without events in the reference (failed alignment / calibration / events-per-base QC, squiggle_read.cpp:320-335) is skipped
        bool have = a.cig_reads[4 * ri + 2] != 0 && rd->n_events > 0 && a.n_pairs[ri] > 0 && !(a.events_per_base[ri] > 5.0) &&
                    (!a.calibrated || a.calibrated[ri] != 0) && first_aligned_read_ge(cv, 0, q_first, r_first) &&
                    last_aligned_read_le_r(cv, max_kmer_idx, q_last, r_last) && q_first <= q_last;
        int first_event = -1, last_event = -1;
        if (have) {
            const int ks = rc ? rl - q_first - k : q_first, ke = rc ? rl - q_last - k : q_last;      // flip_k_strand
            if (ks < 0 || ks >= K || ke < 0 || ke >= K) { have = false; status = NP_EA_BAD_RECORD; }    // the reference asserts / reads out of range
            else { first_event = closest_event(ms, K, ks); last_event = closest_event(ms, K, ke); }
        }
        const bool forward = first_event < last_event;
        int curr_start_event = __builtin_amdgcn_readfirstlane(first_event), curr_start_ref = __builtin_amdgcn_readfirstlane(r_first);

        const float lp_mm_self = rd->trans[0], lp_mb = rd->trans[1], lp_mk = rd->trans[2], lp_mm_next = rd->trans[3],
                    lp_bb = rd->trans[4], lp_bk = rd->trans[5], lp_bm_next = rd->trans[6], lp_bm_self = rd->trans[7],
                    lp_kk = rd->trans[8], lp_km = rd->trans[9];
        const double scale = rd->scale, shift = rd->shift, var = rd->var, log_var = rd->log_var;

        while (have && ((forward && curr_start_event < last_event) || (!forward && curr_start_event > last_event))) {
            // ---- segment geometry (:695-735) ----
            int q_end = 0, r_end = 0;
            if (!last_aligned_ref_le(cv, curr_start_ref + 100, q_end, r_end)) break;      // cannot happen: curr_start_ref is an aligned position
            if (q_end > q_last) { q_end = q_last; r_end = r_last; }
            const bool last_section = q_end == q_last;
            const int curr_end_read = rc ? rl - q_end - k : q_end;
            const int l = r_end - curr_start_ref + 1;
            if (l < 2 * k) break;                                                          // hmm_sequence.length() < 2 * k
            if (curr_start_ref + l > ref_n || curr_end_read < 0 || curr_end_read >= K) { status = NP_EA_BAD_RECORD; break; }
            // (wave-uniform by construction; readfirstlane tells the compiler, so that loop control and addresses stay scalar)
            const int e_start = __builtin_amdgcn_readfirstlane(curr_start_event), e_stop = __builtin_amdgcn_readfirstlane(closest_event(ms, K, curr_end_read));
            const int span = e_start > e_stop ? e_start - e_stop : e_stop - e_start;
            if (span < 2) break;
            const int stride = e_start < e_stop ? 1 : -1;
            const int e = span + 1, n = __builtin_amdgcn_readfirstlane(l - k + 1);
            if (n > NP_EA_MAX_KMERS || e > a.rows_cap) { status = NP_EA_OVERFLOW; break; }
            n_calls++;
            cells += (unsigned long long)(e + 1) * (unsigned long long)(3 * (n + 2)); rows += (unsigned long long)e; kmers += (unsigned long long)n;

            // ---- Viterbi fill (ProfileHMMViterbiOutputR9, r9.inl:130-197): lane owns blocks 2*lane, 2*lane + 1 ----
            np_gauss g[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int b = 2 * lane + c;
                uint32_t rank = 0;
                if (b < n) {
                    // HMMInputSequence::get_kmer_rank(b, k, rc): the forward k-mer at b, or its reverse complement's rank
                    for (int t = 0; t < k; ++t) {
                        const uint32_t code = rc ? 3u - base_code(ref[curr_start_ref + b + k - 1 - t]) : base_code(ref[curr_start_ref + b + t]);
                        rank = rank * 4u + code;
                    }
                }
                g[c] = np_scale_state(a.model, rank, scale, shift, var, log_var);
            }
            const ea_trans tr{lp_mm_self, lp_mb, lp_mk, lp_mm_next, lp_bb, lp_bk, lp_bm_next, lp_bm_self, lp_kk, lp_km};
            const float start_v = ea_fill(g[0], g[1], tr, a.flank[0], ev, bp, e_start, stride, e, n, lane);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);

            // ---- back-track (profile_hmm_align_r9, r9.cpp:117-196): the wave stages NP_EA_CHUNK rows of back-pointers at a
            // time from its scratch into LDS (one coalesced pass) and walks them there.  The walk state is wave-uniform and
            // lives in scalar registers (every lane reads the same LDS byte, readfirstlane makes it a scalar); visited states
            // are collected 64 at a time in a register and flushed with one coalesced store. ----
            int cnt = 0;
            if (start_v != NP_NEG_INF) {                    // assert(get(vm, row, col) != -INFINITY): no path, nothing to emit
                int row = e, kmer = n - 1, ps = 2, stop = 0;
                uint32_t pv = 0;
                while (row > 0 && kmer >= 0 && !stop) {
                    // the line of cell (row, kmer) is row + kmer / 2; along the walk it never grows (each move lowers row or kmer)
                    const int hi = row + (kmer >> 1);
                    const int lo = hi - (NP_EA_CHUNK - 1) > 1 ? hi - (NP_EA_CHUNK - 1) : 1;
                    const int n16 = (hi - lo + 1) * (NP_EA_ROW_BYTES / 16);
                    const uint4* __restrict__ src = (const uint4*)(bp + (size_t)(lo - 1) * NP_EA_ROW_BYTES);
                    for (int i = lane; i < n16; i += 64) ((uint4*)stage)[i] = src[i];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_s_waitcnt(0);
                    __builtin_amdgcn_wave_barrier();
                    const uint8_t* sb = (const uint8_t*)stage;
                    while (row > 0 && kmer >= 0 && row + (kmer >> 1) >= lo) {
                        const uint32_t entry = (uint32_t)row | ((uint32_t)kmer << 16) | ((uint32_t)ps << 24);
                        pv = lane == (cnt & 63) ? entry : pv;
                        cnt++;
                        if ((cnt & 63) == 0) path[cnt - 64 + lane] = pv;
                        const uint32_t byte = (uint32_t)__builtin_amdgcn_readfirstlane((int)sb[(row + (kmer >> 1) - lo) * NP_EA_ROW_BYTES + kmer]);
                        const uint32_t mv = ps == 2 ? (byte & 7u) : ps == 1 ? ((byte >> 3) & 1u) * 2u : ((byte >> 4) == 0u ? 1u : (byte >> 4) == 1u ? 3u : 4u);
                        if (mv == 5u) { stop = 1; break; }          // HMT_FROM_SOFT
                        int next_ps = 2;
                        if (mv == 1u) { kmer -= 1; } else if (mv == 2u) { next_ps = 1; } else if (mv == 3u) { kmer -= 1; next_ps = 1; }
                        else if (mv == 4u) { kmer -= 1; next_ps = 0; }
                        if (ps != 0) row -= 1;                  // K states are silent (r9.cpp:176-178)
                        ps = next_ps;
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                if ((cnt & 63) != 0 && lane < (cnt & 63)) path[(cnt & ~63) + lane] = pv;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_s_waitcnt(0);
            }

            // ---- emit (:774-812): ascending order = the list read backwards ----
            int num_output = 0, last_event_output = 0, last_ref_kmer_output = 0;
            for (int base = 0; base < cnt && (num_output < 50 || last_section); base += 64) {
                const int i = base + lane;
                uint32_t p = 0; bool q = false; int evi = 0, km = 0, ps = 0;
                if (i < cnt) {
                    p = path[cnt - 1 - i];
                    ps = (int)(p >> 24); km = (int)((p >> 16) & 0xff); evi = e_start + ((int)(p & 0xffff) - 1) * stride;
                    q = ps != 0 && evi != curr_start_event;
                }
                const uint64_t qm = __builtin_amdgcn_ballot_w64(q);
                const int before = __builtin_popcountll(qm & ((1ull << lane) - 1ull));
                const int pos = num_output + before;
                const bool wr = q && (pos < 50 || last_section);
                if (wr) {
                    if (n_out + before < out_cap) {
                        a.out_ref[o0 + n_out + before] = curr_start_ref + km;
                        a.out_event[o0 + n_out + before] = evi;
                        a.out_state[o0 + n_out + before] = ps == 2 ? (uint8_t)'M' : (uint8_t)'B';
                    }
                }
                const uint64_t wm = __builtin_amdgcn_ballot_w64(wr);
                const int nw = __builtin_popcountll(wm);
                if (nw > 0) {
                    const int last_lane = 63 - __builtin_clzll(wm);
                    last_event_output = __shfl(evi, last_lane, 64);
                    last_ref_kmer_output = curr_start_ref + __shfl(km, last_lane, 64);
                }
                if (n_out + nw > out_cap) status = NP_EA_OVERFLOW;
                n_out += nw; num_output += nw;
            }
            if (status != NP_EA_OK) break;
            curr_start_event = __builtin_amdgcn_readfirstlane(last_event_output);
            curr_start_ref = __builtin_amdgcn_readfirstlane(last_ref_kmer_output);
            if (num_output == 0) break;
        }
        if (lane == 0) {
            a.n_out[ri] = n_out < out_cap ? n_out : out_cap; a.status[ri] = status; a.n_calls[ri] = n_calls;
            if (a.stats && n_calls > 0) { atomicAdd(a.stats, cells); atomicAdd(a.stats + 1, rows); atomicAdd(a.stats + 2, kmers); }
        }
    }
}

} // namespace

hipError_t np_launch_eventalign_chain(const np_ea_args& a, int n_blocks, hipStream_t s)
{
    hipLaunchKernelGGL(np_eventalign_chain_kernel, dim3(n_blocks), dim3(64), 0, s, a);
    return hipGetLastError();
}
