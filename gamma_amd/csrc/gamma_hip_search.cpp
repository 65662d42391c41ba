// gamma_hip_search.cpp -- the search pipelines of libgamma_hip.so (IVFPQ stages A and B, the small-batch chains, IVFFLAT,
// flat), validity filters, the combining queue of small concurrent calls and the search entry points of the C ABI
// (include/gamma_hip.h).  No CPU fallback: every entry point runs the HIP kernels or returns an error.
#include "gamma_hip_internal.h"
#include "gamma_hip_search.h"

namespace ghi {


// off != nullptr: the range bitmaps go to w_filter at *off (advanced; the caller has sized w_filter for all
// the requests of a combined batch); nullptr: a call of its own, bitmaps from offset 0
// est_codes > 0: about how many list entries (rows, for the flat search) the call will test against the filter; when
// that is several times the number of documents, the field / term clauses are evaluated once per document into a
// bitmap (k_filter_bitmap) and join the request's range bitmaps -- the scan then tests a bit instead of reading column
// values (C5 shape, 10 % range filter on an int64 column: scan 9.6 -> 7.5 ms per 4096 queries)
int build_filter(H* h, const gamma_hip_search_params* p, gh::FilterDesc* f, size_t* off_io, int64_t est_codes) {
    memset(f, 0, sizeof(*f));
    f->del_bitmap = h->d_bitmap;
    f->del_bits = h->d_bitmap ? h->bitmap_bits : 0;
    f->vid2doc = h->h_v2d.empty() ? nullptr : h->d_v2d;
    f->n_vid2doc = (int64_t)h->h_v2d.size();
    f->has_range = p->has_range ? 1 : 0;
    f->n_range = p->has_range ? p->n_range : 0;
    if (f->n_range > gh::kMaxRange) return fail(h, GAMMA_HIP_EINVAL, "too many range filters");
    if (f->n_range > 0) {
        size_t tot = 0;
        for (int i = 0; i < f->n_range; i++) tot += ((size_t)p->range[i].bitmap_bytes + 15) & ~(size_t)15;
        if (!off_io) GH_CHECK(h, h->w_filter.ensure(tot));
        size_t off = off_io ? *off_io : 0;
        for (int i = 0; i < f->n_range; i++) {
            const gamma_hip_range_filter& r = p->range[i];
            uint8_t* dst = h->w_filter.as<uint8_t>() + off;
            GH_CHECK(h, hipMemcpyAsync(dst, r.bitmap, (size_t)r.bitmap_bytes, hipMemcpyHostToDevice,
                                       h->stream));
            f->range[i].bitmap = dst;
            f->range[i].min_doc = r.min_doc;
            f->range[i].max_doc = r.max_doc;
            f->range[i].min_aligned = r.min_aligned;
            f->range[i].b_not_in = r.b_not_in;
            off += ((size_t)r.bitmap_bytes + 15) & ~(size_t)15;
        }
        if (off_io) *off_io = off;
    }
    f->n_field = p->n_field;
    if (p->n_field < 0 || p->n_field > gh::kMaxField || (p->n_field > 0 && !p->field))
        return fail(h, GAMMA_HIP_EINVAL, "bad field filters");
    for (int i = 0; i < p->n_field; i++) {
        const gamma_hip_field_filter& ff = p->field[i];
        auto it = h->fields.find(ff.field_id);
        if (it == h->fields.end()) return fail(h, GAMMA_HIP_EINVAL, "field filter on an unknown column");
        gh::FieldDesc& fd = f->field[i];
        fd.col = it->second.d;
        fd.n = it->second.n;
        fd.dtype = it->second.dtype;
        fd.incl = (ff.include_lower ? 1 : 0) | (ff.include_upper ? 2 : 0);
        fd.lo_i = ff.lower_i;
        fd.hi_i = ff.upper_i;
        fd.lo_f = ff.lower_f;
        fd.hi_f = ff.upper_f;
    }
    f->n_term = p->n_term;
    if (p->n_term < 0 || p->n_term > gh::kMaxTerm || (p->n_term > 0 && !p->term))
        return fail(h, GAMMA_HIP_EINVAL, "bad term filters");
    for (int i = 0; i < p->n_term; i++) {
        const gamma_hip_term_filter& tf = p->term[i];
        auto it = h->terms.find(tf.field_id);
        if (it == h->terms.end()) return fail(h, GAMMA_HIP_EINVAL, "term filter on an unknown column");
        if (tf.n_items < 0 || tf.n_items > gh::kMaxTermItems || tf.op < 0 || tf.op > 2)
            return fail(h, GAMMA_HIP_EINVAL, "bad term filter");
        gh::TermDesc& td = f->term[i];
        td.off = it->second.d_off;
        td.tok = it->second.d_tok;
        td.n = it->second.ndocs;
        td.op = tf.op;
        td.n_items = tf.n_items;
        for (int k = 0; k < tf.n_items; k++) td.items[k] = tf.items[k];
    }
    if (!off_io && est_codes > 0 && f->n_field + f->n_term > 0 && f->n_range < gh::kMaxRange) {
        int64_t nbits = 0;   // beyond every column no clause matches (filter_dev.h), as beyond a range bitmap's max_doc
        for (int i = 0; i < f->n_field; i++) nbits = std::max<int64_t>(nbits, f->field[i].n);
        for (int i = 0; i < f->n_term; i++) nbits = std::max<int64_t>(nbits, f->term[i].n);
        const char* env = getenv("GAMMA_HIP_FILTER_BITMAP");   // 0: never, 1: always (tests), unset: by the estimate
        const bool want = env ? atoi(env) != 0 : est_codes >= 4 * nbits;
        if (want && nbits > 0 && nbits < ((int64_t)1 << 31)) {
            gh::FilterDesc g = *f;   // the clauses alone, on document ids
            g.has_range = 0;
            g.n_range = 0;
            g.del_bitmap = nullptr;
            g.del_bits = 0;
            g.vid2doc = nullptr;
            g.n_vid2doc = 0;
            GH_CHECK(h, h->w_fbits.ensure((size_t)((nbits + 63) / 64) * 8));
            gh::launch_filter_bitmap(h->stream, g, nbits, h->w_fbits.as<uint8_t>());
            gh::RangeDesc& r = f->range[f->n_range++];
            r.bitmap = h->w_fbits.as<uint8_t>();
            r.min_doc = 0;
            r.max_doc = (int32_t)(nbits - 1);
            r.min_aligned = 0;
            r.b_not_in = 0;
            f->has_range = 1;
            f->n_field = 0;
            f->n_term = 0;
        }
    }
    return GAMMA_HIP_OK;
}

int filt_ctx_single(H* h, const gh::FilterDesc& f, FiltCtx* c) {
    GH_CHECK(h, h->w_ftab.ensure(sizeof(gh::FilterDesc)));
    if (!h->ftab_valid || memcmp(&h->ftab_shadow, &f, sizeof(f)) != 0) {
        // the stream may still be reading the previous image: the copy is ordered behind it
        h->ftab_shadow = f;
        h->ftab_valid = true;
        GH_CHECK(h, hipMemcpyAsync(h->w_ftab.p, &h->ftab_shadow, sizeof(f), hipMemcpyHostToDevice, h->stream));
    }
    c->d_tab = h->w_ftab.as<gh::FilterDesc>();
    c->d_qf = nullptr;
    c->any_clause = f.has_range || f.n_field > 0 || f.n_term > 0;
    return GAMMA_HIP_OK;
}

int check_params(H* h, const gamma_hip_search_params* p, int nq, int k) {
    if (!p) return fail(h, GAMMA_HIP_EINVAL, "null params");
    if (nq < 0) return fail(h, GAMMA_HIP_EINVAL, "nq < 0");
    if (p->metric != GAMMA_HIP_METRIC_IP && p->metric != GAMMA_HIP_METRIC_L2)
        return fail(h, GAMMA_HIP_EINVAL, "bad metric");
    if (k > 4096) return fail(h, GAMMA_HIP_EINVAL, "k > 4096 unsupported");
    return GAMMA_HIP_OK;
}

// A request's exact-ties choice (gamma_hip_search_params.exact_ties: 0 = the handle's setting, 1 on, -1 off) is resolved
// ONCE per call, at the entry point, into the call's own copy of the parameter block (exact_ties = 1 / -1); everything
// below reads tie_on(p) -- the handle is not touched.  in_range: the shape is one the replay covers (every recall_num / k
// the ABI accepts; nprobe <= 1024).  Beyond it a request that asked for the mode explicitly gets GAMMA_HIP_EUNSUPPORTED;
// one that relies on the handle's default runs with the (distance, position) order inside ties and is COUNTED
// (gamma_hip_ties_not_honoured) -- never silently.
static inline bool tie_on(const gamma_hip_search_params* p) { return p->exact_ties > 0; }
int resolve_ties(H* h, const gamma_hip_search_params* p, gamma_hip_search_params* pp, bool in_range, const char* what) {
    *pp = *p;
    bool want = p->exact_ties != 0 ? p->exact_ties > 0 : h->exact_ties;
    if (want && !in_range) {
        if (p->exact_ties > 0) return fail(h, GAMMA_HIP_EUNSUPPORTED, what);
        h->ties_unhonoured.fetch_add(1, std::memory_order_relaxed);
        want = false;
    }
    pp->exact_ties = want ? 1 : -1;
    return GAMMA_HIP_OK;
}

// ---- IVFPQ stage A: coarse + tables + scan + top-R + ids ------------------------------
// results: w_cand_dis [nq*R] (ADC distance, best first, sentinel pad), w_cand_ids [nq*R]
// the assignment is complete on the stream (the heap replay of tied rows may still be running on the side stream)
int coarse_join(H* h) {
    if (h->coarse_join_pending) {
        GH_CHECK(h, hipStreamWaitEvent(h->stream, h->ev_join, 0));
        h->coarse_join_pending = false;
    }
    return GAMMA_HIP_OK;
}

// defer_join: the caller has kernels to launch that do not read the assignment and calls coarse_join itself
int ivfpq_coarse(H* h, const gamma_hip_search_params* p, int nq, const float* d_x, float* out_dis = nullptr,
                 int* out_probe = nullptr, bool defer_join = false) {
    const int P = p->nprobe, d = h->d, nlist = h->nlist;
    hipStream_t s = h->stream;
    int mode = p->coarse_mode;
    if (mode < 0) mode = nq < 20 ? 0 : 1;  // faiss:utils/distances.cpp:303,346
    // large batches: no distance matrix (coarse.hip); with exact ties its rows with a tie near the cut are recomputed
    // and replayed through the reference's heap by the repair kernel
    const bool fused = mode == 1 && h->coarse_fused && gh::coarse_fused_supported(nq, d, nlist, P, tie_on(p));
    gh::CoarseFusedPlan plan;
    if (fused) {
        plan = gh::coarse_fused_plan(nq, nlist, P, h->coarse_cap, tie_on(p));
        GH_CHECK(h, h->w_mat.ensure(plan.bytes));
    } else {
        GH_CHECK(h, h->w_mat.ensure((size_t)nq * nlist * sizeof(float)));
    }
    if (!out_dis || !out_probe) {   // the workspace the scan reads
        GH_CHECK(h, h->w_coarse_dis.ensure((size_t)nq * P * sizeof(float)));
        GH_CHECK(h, h->w_probe.ensure((size_t)nq * P * sizeof(int)));
        out_dis = h->w_coarse_dis.as<float>();
        out_probe = h->w_probe.as<int>();
    }
    StageScope t(h, GAMMA_HIP_STAGE_COARSE);
    if (fused) {
        static const bool no_side = getenv("GAMMA_HIP_NO_SIDE_STREAM") != nullptr;
        const bool side = tie_on(p) && defer_join && !no_side;
        gh::launch_coarse_fused(s, plan, h->w_mat.p, d_x, nq, d, h->d_cc, nlist, h->d_cc_norms, P, out_dis, out_probe,
                                tie_on(p), h->d_tie_stats, side ? h->side : nullptr, h->ev_fork, h->ev_join);
        h->coarse_join_pending = side;
        static const bool dbg = getenv("GAMMA_HIP_COARSE_DBG") != nullptr;
        if (dbg) {   // how many queries the strip lists could not hold (they went through the repair kernel)
            int n_ovf = 0;
            GH_CHECK(h, hipStreamSynchronize(s));
            GH_CHECK(h, hipMemcpy(&n_ovf, static_cast<char*>(h->w_mat.p) + plan.off_ovf, sizeof(int), hipMemcpyDeviceToHost));
            fprintf(stderr, "coarse fused: nlist %d nprobe %d sample %d strips %d: %d of %d queries repaired\n", nlist, P,
                    plan.sample, plan.nseg, n_ovf, nq);
        }
        return GAMMA_HIP_OK;
    }
    if (mode == 0) {
        gh::launch_pairwise(s, true, d_x, nq, d, h->d_cc, nlist, h->w_mat.as<float>(), nlist);
    } else {
        // query norms: fused into the MFMA kernel (xn = nullptr) when a tile holds whole rows (d <= 128); longer rows
        // get them from their own pass -- inside the K-slab loop they cost a fifth of the kernel (d = 768: 3.46 -> 2.72 ms
        // per 8192 x 16384 with the conflict-free staging)
        const float* xn = nullptr;
        if (d > 128) {
            GH_CHECK(h, h->w_xn.ensure((size_t)nq * sizeof(float)));
            gh::launch_row_norms(s, d_x, nq, d, h->w_xn.as<float>());
            xn = h->w_xn.as<float>();
        }
        gh::launch_l2_gemmform(s, d_x, nq, d, h->d_cc, nlist, xn, h->d_cc_norms,
                               h->w_mat.as<float>(), nlist, true);
    }
    if (tie_on(p)) GH_CHECK(h, h->w_tieflag.ensure((size_t)nq));
    static const bool no_side = getenv("GAMMA_HIP_NO_SIDE_STREAM") != nullptr;
    const bool side = tie_on(p) && defer_join && !no_side;
    h->coarse_join_pending = gh::launch_coarse_select(s, h->w_mat.as<float>(), nlist, nq, P, out_dis, out_probe,
                                                      tie_on(p) ? h->w_tieflag.as<uint8_t>() : nullptr, h->d_tie_stats,
                                                      side ? h->side : nullptr, h->ev_fork, h->ev_join);
    return GAMMA_HIP_OK;
}

// pre_dis / pre_probe: coarse assignment computed elsewhere (sharded search: the rank owning the
// query slice), device pointers [nq*nprobe]; nullptr = run the coarse quantizer here
// Two-phase list-shard search (round 6): between the producers (the query's nearest probes OWNED BY THIS SHARD: they
// bound the shard's recall_num-th best) and the consumers (all its other probes) the bounds of all shards are reduced
// -- min for L2, max for inner product: one float per query -- so that every shard filters against a bound of the GLOBAL
// recall_num-th best instead of its own, W times looser one.  `reduce` is the caller's collective (RCCL all-reduce in
// gamma_amd/dist.py, peer reads in the in-process group); it is called exactly ONCE per shard call, with neutral values
// when this call cannot run in two phases.
struct BoundXchg {
    float* d_bound = nullptr;                  // [nq of the whole call] the caller's buffer; holds the reduced bounds afterwards
    gamma_hip_bound_reduce_fn fn = nullptr;
    void* user = nullptr;
    bool called = false;
    bool two_phase = false;                    // this call runs in two phases (one chunk, bounded scan possible)
    int P_in = 0;                              // row length of the supplied assignment (p->nprobe is the compacted rows')
    int G1 = 2;                                // probes of the producers' group
    int reduce(H* h, int nq, bool l2) {
        called = true;
        const int rc = fn ? fn(user, d_bound, nq, l2 ? 0 : 1, (void*)h->stream) : 0;
        return rc ? fail(h, GAMMA_HIP_EDEVICE, "bound reduction callback failed") : GAMMA_HIP_OK;
    }
};

int ivfpq_stage_a(H* h, const gamma_hip_search_params* p, const FiltCtx& fc, int nq,
                  const float* d_x, int R, const float* pre_dis = nullptr, const int* pre_probe = nullptr,
                  bool shard = false, float* out_dis = nullptr, int64_t* out_ids = nullptr, BoundXchg* bx = nullptr) {
    const int P = p->nprobe, d = h->d, M = h->M, nlist = h->nlist;
    const bool two = bx && bx->two_phase;
    const bool l2 = p->metric == GAMMA_HIP_METRIC_L2;
    hipStream_t s = h->stream;
    // this call reads the lists through the version of their (offset, length) tables that is current now:
    // behind the writer's copies (ver_ev), and the version is not reused before the kernels below are done (rd_ev)
    const int ver = h->cur_ver;
    GH_CHECK(h, hipStreamWaitEvent(s, h->ver_ev[ver], 0));
    GH_CHECK(h, h->w_coarse_dis.ensure((size_t)nq * P * sizeof(float)));
    GH_CHECK(h, h->w_probe.ensure((size_t)nq * P * sizeof(int)));
    GH_CHECK(h, h->w_pair_off.ensure((size_t)nq * (P + 1) * sizeof(int)));
    GH_CHECK(h, h->w_pair_base.ensure((size_t)nq * P * sizeof(int64_t)));
    GH_CHECK(h, h->w_qtotal.ensure((size_t)nq * sizeof(int)));
    GH_CHECK(h, h->w_cand_pos.ensure((size_t)nq * R * sizeof(int)));
    // the top-R table goes to the workspace (stage B reads it there) or straight into the caller's
    // buffers (sharded search: 12 B x R per query would otherwise be copied once more)
    if (!out_dis) {
        GH_CHECK(h, h->w_cand_dis.ensure((size_t)nq * R * sizeof(float)));
        out_dis = h->w_cand_dis.as<float>();
    }
    if (!out_ids) {
        GH_CHECK(h, h->w_cand_ids.ensure((size_t)nq * R * sizeof(int64_t)));
        out_ids = h->w_cand_ids.as<int64_t>();
    }
    if (pre_dis && pre_probe) {
        if (shard) {
            // dense probe groups for the owned lists (kernels.hip, k_compact_probes)
            gh::launch_compact_probes(s, pre_probe, pre_dis, nq, two ? bx->P_in : P, h->d_list_len, h->d_list_mask, nlist,
                                      h->w_probe.as<int>(), h->w_coarse_dis.as<float>(), P);
        } else {
            GH_CHECK(h, hipMemcpyAsync(h->w_coarse_dis.p, pre_dis, (size_t)nq * P * sizeof(float),
                                       hipMemcpyDeviceToDevice, s));
            GH_CHECK(h, hipMemcpyAsync(h->w_probe.p, pre_probe, (size_t)nq * P * sizeof(int),
                                       hipMemcpyDeviceToDevice, s));
        }
    } else {
        GH_TRY(ivfpq_coarse(h, p, nq, d_x, nullptr, nullptr, /*defer_join=*/true));
    }
    h->scan_pairs += (int64_t)nq * P;
    // ids are only read during the scan when something can reject an entry: a delete bit,
    // a range filter, or a superseded (bit 63) slot left behind by Update
    const int need_ids =
            (!h->prefiltered && (fc.any_clause || (h->d_bitmap && h->bitmap_any) || h->n_moved > 0)) ? 1 : 0;
    const int* qperm = nullptr;
    // Probes per workgroup.  Sharded search with a compacted assignment: a query keeps ~P/W probes on
    // this shard, all in its first group(s) -- the other P/G - 1 workgroups of the query would start only
    // to find nothing to do (at W = 8 that was half of the scan time).  When the expected candidate
    // count per query is small, ONE workgroup takes all of a query's probes (G = P): it bounds the
    // R-th best itself (producer path of the pre-filter, no consumers), and computes the query's PQ
    // table on the fly instead of reading it back from HBM (IPF, kernels.hip).
    // 8 probes per workgroup pay off with short lists (half the query-table re-reads, a tighter bound from a
    // first group of 8 lists); long lists or many probes balance better with 4 (tools/shape_sweep.py)
    // codes per list of the lists THIS handle scans: a list shard holds (or, under a mask, scans) 1 / W of the lists
    double mean_len = (double)h->ntotal / std::max(1, nlist);
    int64_t owned = nlist;
    if (shard && !h->h_list_mask.empty()) {
        int64_t tot = 0;
        owned = 0;
        for (int l = 0; l < nlist; l++)
            if (h->h_list_len[l] > 0 && h->h_list_mask[l]) {
                owned++;
                tot += h->h_list_len[l];
            }
        mean_len = owned ? (double)tot / (double)owned : 0.0;
    }
    int G0 = (mean_len <= 700.0 && P <= 64) ? 8 : 4;
    // (with the byte-table filter pass a consumer probe costs a third of a producer's: five probes bound nearly as well as
    //  eight -- C3: scan 729 us at 8, 691 at 6, 676 at 5, 671 at 4 but with queries whose slices overflow; GAMMA_HIP_SCAN_G to sweep)
    static const bool no_c8 = getenv("GAMMA_HIP_NO_C8") != nullptr;
    static const bool c8_m32 = getenv("GAMMA_HIP_C8_M32") != nullptr;
    // (recall_num up to 512 since round 6 -- with the wave-per-query selection of eight sorted runs, select.hip: C3 at
    //  recall_num 300 3.00 -> 2.07 ms per 16384 queries (scan 0.475 -> 0.675 of the roofline, select 910 -> 277 us), the C4 shape
    //  at 8 M 5.19 -> 3.96 ms per 8192; the candidate stages (1536 / 768 slots) overflow into the unfiltered path as ever)
    static const int cf_maxr = getenv("GAMMA_HIP_CF_MAXR") ? atoi(getenv("GAMMA_HIP_CF_MAXR")) : 512;
    const bool c8_shape = !no_c8 && l2 && (M == 16 || (M == 32 && c8_m32)) && R <= cf_maxr && h->d_sums && h->d_t2max;
    if (G0 == 8 && c8_shape && P > 8) G0 = 5;
    int64_t t2_bytes = (int64_t)nlist * M * 256 * sizeof(float);
    const bool compacted = shard && pre_dis && pre_probe;
    if (two) {
        // the producers take the query's nearest G1 owned probes only: everything else waits for the GLOBAL bound
        G0 = std::max(1, std::min(bx->G1, P));
        t2_bytes = owned * M * 256 * (int64_t)sizeof(float);
    } else if (compacted && h->scan_bound && R <= 256) {
        if (h->h_list_mask.empty()) {
            owned = 0;
            for (int l = 0; l < nlist; l++) owned += h->h_list_len[l] > 0;
        }
        t2_bytes = owned * M * 256 * (int64_t)sizeof(float);
        const double exp_probes = (double)P * (double)owned / std::max(1, nlist);
        const double exp_cand = exp_probes * mean_len;
        // (not beyond that: one workgroup walking 50 k - 200 k codes of a query is the long pole of the launch -- full-size
        //  C4 emulated, profiles/r04_scaling_emul.txt: W = 2 25.4 ms per step with 4 or 8 probes per workgroup, 27.9 with
        //  16, 31.4 with 32 or 64; W = 4 27.0 / 27.1 / 30.1 / 30.3 / 30.2)
        if (exp_cand <= 16384.0) {
            G0 = 1;
            while (G0 < P) G0 <<= 1;
        }
    }
    const int G = gh::scan_group_size(nq, P, G0), PGN = (P + G - 1) / G;
    // exact ties (ties.hip): queries whose top-R cut goes through a group of equal ADC distances are marked here
    // and redone by the replay at the end of stage B
    h->tie = H::TieCtx();
    // (a shard marks the cut ties of its own top-R too: the merge at the slice's owner asks for them,
    //  gamma_hip_ivfpq_shard_cut_flags)
    h->tie.on = tie_on(p);
    h->shard_cut_nq = (shard && h->tie.on) ? nq : 0;
    h->shard_cut_chunked = false;
    // Threshold pre-filter: scan the nearest probe group first, bound each query's R-th best
    // distance from it, and let the scan of the remaining groups keep a short survivor list per
    // query; the exact top-R then comes from a few hundred survivors instead of ~10^4 candidates
    // (select.hip).  Queries without a usable bound fall back to the unfiltered selection.
    // (sharded without a supplied assignment: probe groups are sparse, nothing to bound from)
    // small batches: one probe per workgroup, a single list rarely holds R candidates, and the
    // unfiltered selection is latency-bound anyway
    // (one group per query -- few probes, or a shard -- is fine: the producer bounds and compacts its own candidates)
    // (recall_num up to 1024 since round 5: slices of 2048 items and k_select_final_wg beyond 256 -- the configurations that need
    //  a long short-list, full-size C5 at ~1000, keep the pre-filter; GAMMA_HIP_BOUND_MAXR: the old gate for A/B runs)
    static const int scan_gmin = getenv("GAMMA_HIP_SCAN_GMIN") ? atoi(getenv("GAMMA_HIP_SCAN_GMIN")) : 4;
    static const int bound_maxr = getenv("GAMMA_HIP_BOUND_MAXR") ? atoi(getenv("GAMMA_HIP_BOUND_MAXR")) : 1024;
    bool bounded = (!shard || compacted) && h->scan_bound && R <= std::min(1024, bound_maxr) && P <= 128 && G >= (two ? 1 : scan_gmin);
    if (bounded) {
        // feedback (gamma_hip_internal.h, bound_*): the counts of some recent call are in the pinned words
        static const bool no_fb = getenv("GAMMA_HIP_NO_BOUND_FEEDBACK") != nullptr;
        // (what was learnt holds for one kind of call: other cuts or a filter bound differently -- start over, with
        //  counters of its own: the two slots alternate, so counts of the previous kind still on their way land elsewhere)
        const uint64_t sig = ((uint64_t)P << 40) ^ ((uint64_t)R << 20) ^ ((uint64_t)(l2 ? 1 : 0) << 1) ^ (fc.any_clause ? 1u : 0u) ^
                             ((uint64_t)(shard ? 1 : 0) << 2) ^ (1ull << 63);
        if (sig != h->bound_sig) {
            h->bound_sig = sig;
            h->bound_epoch++;
            h->bound_off_calls = 0;
            h->bound_calls = 0;
            h->bound_seen[0] = h->bound_seen[1] = 0;
            const int slot = h->bound_epoch & 1;
            if (h->bound_copy_pending) {   // a copy of the previous kind's counts still on its way: it lands before the words are reused
                (void)hipEventSynchronize(h->bound_copy_ev);
                h->bound_copy_pending = false;
            }
            h->pin_bound_stat[2 * slot] = h->pin_bound_stat[2 * slot + 1] = 0;
            GH_CHECK(h, hipMemsetAsync(h->d_bound_stat + 2 * slot, 0, 2 * sizeof(unsigned long long), s));
        }
        const int slot = h->bound_epoch & 1;
        if (no_fb || h->bound_feedback_off) {
        } else if (h->bound_off_calls > 0) {
            if (--h->bound_off_calls > 0) bounded = false;   // (0: this call re-probes)
        } else {
            // (the pair is consulted only once the copy that writes it has completed -- bound_copy_ev -- so the two
            //  words belong to one state of the counters; a sample that runs backwards, e.g. a late copy landing on words
            //  zeroed for a new kind of call, is ignored, and the fell-through count can never exceed the queries)
            const bool landed = !h->bound_copy_pending || hipEventQuery(h->bound_copy_ev) == hipSuccess;
            if (landed) h->bound_copy_pending = false;
            else (void)hipGetLastError();
            const unsigned long long u = h->pin_bound_stat[2 * slot], n = h->pin_bound_stat[2 * slot + 1];
            const unsigned long long dn = n - h->bound_seen[1];
            const unsigned long long du = std::min(u >= h->bound_seen[0] ? u - h->bound_seen[0] : 0ull, dn);
            if (landed && n >= h->bound_seen[1] && u >= h->bound_seen[0] && dn >= 2048) {
                h->bound_seen[0] = u;
                h->bound_seen[1] = n;
                if (2 * du > dn) {   // more than half of the recent queries went to the unfiltered selection anyway
                    h->bound_off_calls = 256;
                    h->bound_backoffs++;
                    bounded = false;
                }
            }
        }
    }
    // L2 table mode 0 (no precomputed table, H::table_mode): per-list tables from the residual (scan.hip RES) -- the plain and
    // the bounded loop; no query table, none of the passes built on T2 sums (d_sums is null: cf / c8 / q8 are off by their own gates)
    const bool res = l2 && h->table_mode == 0;
    const bool fuse_ip = bounded && !res && PGN == 1 && (M == 16 || M == 32) && !getenv("GAMMA_HIP_NO_FUSED_IP");
    // the bounded scan's per-call state (repair list, ready words, survivor counts) is sized here, before the pair offsets,
    // whose kernel clears it together with the tie flags -- one launch instead of four fills in front of the scan
    // (byte-table filter pass: a first group of a few probes bounds loosely for some queries -- slices of 2048 keep them out of the
    //  unfiltered path; GAMMA_HIP_SLICE_CAP to sweep)
    static const int cap_env = getenv("GAMMA_HIP_SLICE_CAP") ? atoi(getenv("GAMMA_HIP_SLICE_CAP")) : 0;
    // (clamped to 2048: the selection and the tie replay hold one slice in LDS)
    const int cap = cap_env > 0 ? std::min(2048, std::max(cap_env, gh::scan_slice_cap(R))) : gh::scan_slice_cap(R);
    bool cf_ok = false, q8_ok = false, q8_fused = false, one_wg = false;
    int PGM = PGN, nsl = PGN, cf_span = 0;
    unsigned long long* ready = nullptr;
    if (bounded) {
        // filter pass of the consumers (kernels.hip, CF): needs the sums beside the arena the scan reads -- not the
        // shadow arena of a call running over lists compacted under its filter.  With it ONE consumer workgroup per
        // query takes every probe behind the producer's: two groups, two slices per query.
        static const bool no_cf = getenv("GAMMA_HIP_NO_SCAN_CF") != nullptr;
        // (short lists only: with thousands of codes per list the per-pair table costs next to nothing, while ONE consumer
        //  workgroup per query overruns its candidate stage and sends the query to the unfiltered path -- full-size C4, 6100
        //  codes per list: 38.6 ms per 8192 queries with the filter pass, 27.6 without)
        // Short lists only (mean <= 2000 codes).  With thousands of codes per list the per-pair table costs next to
        // nothing and the pass loses: full-size C4 (6100 codes per list, 8192 queries) 27.7 ms without it; 38.6 ms with ONE
        // consumer workgroup per query (it stages more candidates than its 768 slots hold and the query goes to the
        // unfiltered path); 32.5 ms with the probes behind the producer's cut into consumer groups of ~48 k codes, each
        // with its own stage and slice (round 4, profiles/r04_scale_runs.txt) -- the margin candidates' exact recompute
        // and the second look at the codes cost more than the table build they save.  The split (cf_span) stays for the
        // lists in between: a consumer group takes at most ~64 k codes.
        static const double cf_maxlen = getenv("GAMMA_HIP_SCAN_CF_MAXLEN") ? atof(getenv("GAMMA_HIP_SCAN_CF_MAXLEN")) : 2000.0;
        static const double cf_codes = getenv("GAMMA_HIP_SCAN_CF_CODES") ? atof(getenv("GAMMA_HIP_SCAN_CF_CODES")) : 65536.0;
        // (and short-lists only: the pass stages SCAN_CF_CAP = 768 candidates per consumer workgroup)
        cf_ok = !no_cf && R <= cf_maxr && (!h->prefiltered || h->cmp_has_sums) && PGN > 1 && mean_len <= cf_maxlen &&
                gh::scan_cf_applies(l2, M, P, G, h->d_sums && h->d_t2max, false);
        // list-major byte-table pass for the consumer probes (q8scan.hip, round 5): same conditions as the filter pass it
        // replaces, one validity predicate for the whole call, not a shard (GAMMA_HIP_NO_Q8: the query-major pass, for A/B)
        static const bool no_q8 = getenv("GAMMA_HIP_NO_Q8") != nullptr;
        static const double q8_maxlen = getenv("GAMMA_HIP_Q8_MAXLEN") ? atof(getenv("GAMMA_HIP_Q8_MAXLEN")) : 1e12;
        // (long lists: a tile's table staging and its chain of dependent loads are amortised over thousands of codes; at C3's
        //  244 codes per list a tile is one step of 64 codes per wave and the pass is latency-bound -- measured slower than the
        //  query-major pass there, profiles/r05_q8_*.txt -- so short lists keep the query-major filter pass)
        // (the switch, measured after the byte image went into the query-major pass: nlist 4096 / M 16 / nprobe 32 -- 732 codes per
        //  list 2.81 ms query-major, 2.84 list-major; 976: 3.31 / 3.20; 1465: 4.49 / 3.57 -- nlist 16384 / M 32 / nprobe 64 -- 732: 4.35 /
        //  4.38; 976: 5.36 / 4.87)
        static const double q8_minlen = getenv("GAMMA_HIP_Q8_MINLEN") ? atof(getenv("GAMMA_HIP_Q8_MINLEN")) : 800.0;
        const int64_t q_stride0 = (std::max<int64_t>(1, (int64_t)P * std::max(1, h->max_list_len)) + 3) & ~(int64_t)3;
        // (a list shard with a supplied, compacted assignment runs it too: pairs of lists of other shards are simply not placed)
        q8_ok = !no_q8 && !no_cf && R <= 1024 && (!shard || compacted) &&
                // (shadow lists of the standing deletes are as long as the lists: the pass stays on over them -- C4 shape at 20 M with
                //  5 % deleted: 6.2 ms per 8192 queries without it, 4.5 unfiltered; lists cut down under a request's own clause
                //  are as short as the clause makes them: the query-major pass)
                (!h->prefiltered || (h->cmp_has_sums && !h->cmp_by_clause)) && !fc.d_qf && PGN > 1 && mean_len <= q8_maxlen &&
                gh::scan_cf_applies(l2, M, P, G, h->d_sums && h->d_t2max, false) && gh::q8_supported(M, P, G, q_stride0) && mean_len >= q8_minlen && nlist <= 16384;
        if (q8_ok) cf_ok = false;
        if (q8_ok) {
            PGM = 1;     // the main launch: producers only
            nsl = PGN;   // slice 0: the producer's, slices 1 ..: the consumer probe groups' (k_q8_exact)
        } else if (cf_ok) {
            const int rest = P - G;
            int nc = (int)std::ceil(rest * mean_len / cf_codes);
            nc = std::max(1, std::min(nc, std::max(1, PGN - 1)));
            cf_span = nc > 1 ? (rest + nc - 1) / nc : 0;
            PGM = 1 + (nc > 1 ? (rest + cf_span - 1) / cf_span : 1);
            // ONE workgroup per query (scan.hip, ScanBound::prod_c8): the bound from byte-image estimates of the first group, then
            // the filter pass over all probes in the same workgroup; slice 0 = the survivors, slice 1 = those of the second stage
            // of a query whose first group took the regular producer
            // OFF by default (GAMMA_HIP_PROD_C8=1, parity-tested): the image's PROVEN error width (2.03 M delta: all sixteen
            // quantisation errors of a code aligned) is ~14 times the typical error, the bound lets 875 survivors per query through
            // instead of 225 and most queries overflow their slice (C3: scan 958 us, select 672 us against 613 / 101)
            static const bool prod_c8_on = getenv("GAMMA_HIP_PROD_C8") != nullptr && atoi(getenv("GAMMA_HIP_PROD_C8")) != 0;
            // (no validity predicates on this path: under a filter the first group rarely holds recall_num valid codes)
            one_wg = prod_c8_on && c8_shape && M == 16 && !shard && !two && cf_span == 0 && P <= 64 && !need_ids;
            if (one_wg) PGM = 1;
        } else {
            PGM = PGN;   // probe groups of the main launch
        }
        if (!q8_ok) nsl = one_wg ? 2 : PGM;   // one survivor slice per probe group (slice 0: the producer's own)
        // rq | ready[nq] | gcnt[nq][nsl]   (rq: count + list of the queries that need the repair launch, 8-byte aligned)
        const size_t rq_bytes = (((size_t)nq + 1) * sizeof(int) + 7) & ~(size_t)7;
        GH_CHECK(h, h->w_scnt.ensure(rq_bytes + (size_t)nq * (sizeof(unsigned long long) + (size_t)nsl * sizeof(int))));
        GH_CHECK(h, h->w_sflag.ensure((size_t)nq));
        GH_CHECK(h, h->w_surv.ensure((size_t)nq * nsl * cap * sizeof(unsigned long long)));
        ready = reinterpret_cast<unsigned long long*>(h->w_scnt.as<char>() + rq_bytes);
    }
    {
        StageScope t(h, GAMMA_HIP_STAGE_TABLES);
        // two-phase list shard on the list-major pass: every consumer of the query tables computes them on the fly (the
        // producers: IPF; k_q8_quant / k_q8_exact: fx) -- W x 32 KB per query through HBM otherwise; only the rows of the
        // queries the repair launch re-scores are written (launch_pq_ip_table_rows below)
        // MEASURED SLOWER and off by default (GAMMA_HIP_Q8_FUSED_IP=1): an entry of the table is 4 bytes, the dsub = 4 codebook
        // row it is made from 16 -- every workgroup pulling the 128 KB codebook through the L2 costs more than the 32 KB table
        // from HBM (one emulated rank of 8, C4 shape 20 M: tables 0.49 -> 0.05 ms but scan 4.35 -> 5.5).  What does pay is
        // k_q8_exact computing only its candidates' entries (Q8Args::xd).
        static const bool q8_fused_on = getenv("GAMMA_HIP_Q8_FUSED_IP") != nullptr;
        q8_fused = two && q8_ok && q8_fused_on && (M == 16 || M == 32);
        if (!fuse_ip && !res) {
            GH_CHECK(h, h->w_st2.ensure((size_t)nq * M * 256 * sizeof(float)));
            if (!q8_fused) gh::launch_pq_ip_table(s, d_x, nq, d, M, h->d_pqc, h->w_st2.as<float>());
        }
        GH_TRY(coarse_join(h));
        // a deferred replay of the previous call / chunk reads what is written from here on (flag lists, pair offsets,
        // slab, survivor slices, candidate tables): it has had the coarse quantizer and the tables to finish behind
        GH_TRY(replay_join(h));
        if (h->tie.on) {
            GH_CHECK(h, h->w_tcut.ensure((size_t)nq));
            GH_CHECK(h, h->w_tlist.ensure(((size_t)nq + 1) * sizeof(int)));   // count | list[nq]
        }
        gh::PairZero pz;
        // enough queries that L2 capacity matters: run them in spatial order (kernels.hip)
        // (not when the T2 rows of the lists scanned here fit the L2s anyway: a shard of a small index)
        const bool order = h->sort_queries && h->d_list_rank && nq >= 256 && t2_bytes > ((int64_t)8 << 20);
        int* qo_bins = nullptr;
        if (order) {
            GH_CHECK(h, h->w_qperm.ensure((size_t)2 * nq * sizeof(int)));   // qperm | qkey
            if (gh::query_order_grid(nq)) {
                if (!h->w_qbins.p) {   // histogram | cursors; every call leaves the histogram zero
                    GH_CHECK(h, h->w_qbins.ensure((size_t)2 * gh::query_order_bins() * sizeof(int)));
                    GH_CHECK(h, hipMemsetAsync(h->w_qbins.p, 0, (size_t)2 * gh::query_order_bins() * sizeof(int), s));
                }
                qo_bins = h->w_qbins.as<int>();
                pz.qo_rank = h->d_list_rank;
                pz.qo_key = h->w_qperm.as<int>() + nq;
                pz.qo_bins = qo_bins;
            }
        }
        if (h->tie.on) {
            pz.bytes = h->w_tcut.as<uint8_t>();
            pz.count_a = h->w_tlist.as<int>();
        }
        if (bounded) {
#ifdef GH_SCAN_TIMING   // timing experiments with INVALID results: only in a library built for them (-DGH_SCAN_TIMING)
            static const int scan_dbg_part = getenv("GAMMA_HIP_SCAN_PART") ? atoi(getenv("GAMMA_HIP_SCAN_PART")) : 0;
            // (timing experiment 3: every other call runs the consumers alone, on the bounds of the call before)
            static std::atomic<int> scan_dbg_calls{0};
            h->scan_dbg_now = scan_dbg_part == 3 ? ((scan_dbg_calls.fetch_add(1) & 1) ? 2 : 0) : scan_dbg_part;
#endif
            pz.words = h->scan_dbg_now == 2 ? nullptr : ready;
            pz.count_b = h->w_scnt.as<int>();
        }
        gh::launch_pair_offsets(s, h->w_probe.as<int>(), nq, P, h->d_list_len, h->d_list_mask, nlist,
                                h->w_pair_off.as<int>(), h->w_qtotal.as<int>(),
                                h->profile == 1 ? h->d_scan_codes : nullptr, h->d_list_off,
                                h->w_pair_base.as<int64_t>(), &pz);
        if (order) {
            gh::launch_query_order(s, h->w_probe.as<int>(), nq, P, h->d_list_rank, nlist,
                                   h->w_qperm.as<int>() + nq, h->w_qperm.as<int>(), qo_bins, qo_bins != nullptr);
            qperm = h->w_qperm.as<int>();
        }
        h->last_qperm = qperm;   // stage B runs the re-rank in the same order
    }
    // per-query slab of the distance buffer; multiple of 4 floats so rows are 16-byte aligned
    int64_t q_stride = (std::max<int64_t>(1, (int64_t)P * std::max(1, h->max_list_len)) + 3) & ~(int64_t)3;
    // (the longest candidate row of THIS batch, measured on its assignment: list shards, and whole-index calls whose general
    //  stride -- nprobe x the longest list -- would cut the batch into chunks)
    if (h->q_stride_cap > 0) q_stride = std::min(q_stride, h->q_stride_cap);
    GH_CHECK(h, h->w_dist.ensure((size_t)nq * q_stride * sizeof(float)));
    // dis0 of every (query, probe) pair: the coarse distance (L2) or <x_q, centroid> (inner product)
    const float* dis0 = h->w_coarse_dis.as<float>();
    if (!l2) {
        StageScope t(h, GAMMA_HIP_STAGE_TABLES, false);
        GH_CHECK(h, h->w_pair_ip.ensure((size_t)nq * P * sizeof(float)));
        gh::launch_pair_ip(s, d_x, h->d_cc, h->w_probe.as<int>(), nq, P, d, nlist, h->w_pair_ip.as<float>());
        dis0 = h->w_pair_ip.as<float>();
    }
    auto scan = [&](int gsz, int pg_lo, int pg_cnt, const gh::ScanBound* bound, bool count) {
        StageScope t(h, GAMMA_HIP_STAGE_SCAN, count);
        gh::launch_ivfpq_scan_pair(s, l2, d_x, nq, d, M, P, h->w_probe.as<int>(),
                                   dis0, h->d_cc, h->scan_st2(l2), h->d_T2,
                                   h->d_list_off, h->d_list_len, h->d_list_mask, nlist, h->d_codes,
                                   h->d_ids, h->w_pair_off.as<int>(), q_stride, h->w_dist.as<float>(),
                                   fc.d_tab, fc.d_qf, need_ids, qperm, gsz, pg_lo, pg_cnt, shard ? 1 : 0, bound,
                                   fuse_ip ? h->d_pqc : nullptr);
    };
    if (!bounded) {
        scan(G, 0, PGN, nullptr, true);
        StageScope t(h, GAMMA_HIP_STAGE_SELECT);
        gh::launch_select_topk(s, l2, h->w_dist.as<float>(), q_stride, h->w_qtotal.as<int>(), 0,
                               (int)std::min<int64_t>(q_stride, 1 << 30), nq, R,
                               out_dis, h->w_cand_pos.as<int>());
        gh::launch_map_candidates(s, h->w_cand_pos.as<int>(), nq, R, P, h->w_probe.as<int>(),
                                  h->w_pair_off.as<int>(), h->d_list_off, h->d_ids,
                                  out_ids);
        if (h->tie.on)
            gh::launch_flag_cut_ties(s, h->w_dist.as<float>(), q_stride, h->w_qtotal.as<int>(), nq, R, out_dis,
                                     h->w_cand_pos.as<int>(), nullptr, h->w_tcut.as<uint8_t>());
    } else {
        gh::ScanBound sb = {};   // (every field starts at 0: part, dbg_part, c8 ...)
        sb.ready = ready;
        sb.surv = h->w_surv.as<unsigned long long>();
        sb.gcnt = reinterpret_cast<int*>(ready + nq);
        sb.K = R;
        sb.cnt_stride = nsl;
        // consumer groups with a bound keep only their survivors; what the unfiltered fallback and the tie replay of a
        // query without a usable bound need is stored by the repair launch below
        sb.store_all = 0;
        sb.rq_count = h->w_scnt.as<int>();
        sb.rq_list = h->w_scnt.as<int>() + 1;
        sb.sums = cf_ok ? h->d_sums : nullptr;
        sb.t2max = cf_ok ? h->d_t2max : nullptr;
        sb.t2max_all = h->t2max_all;
        sb.pair_base = h->w_pair_base.as<int64_t>();
        sb.cf_span = cf_ok ? cf_span : 0;
        static const int spins_env = getenv("GAMMA_HIP_SCAN_SPINS") ? atoi(getenv("GAMMA_HIP_SCAN_SPINS")) : 0;
        sb.spins = spins_env;
        sb.timeouts = h->d_bound_stat + 4;
        sb.slice_cap = cap;
        // the producer on the filter pass's arithmetic too (scan.hip, prod_cf; with the query-major filter pass, not for shards).
        // OFF by default: measured slower at C3 -- scan 828 us against 784 (878 before the variant was held to six waves per
        // SIMD: 89 VGPRs), + 25 us for re-scoring the first group of the tie-flagged queries in front of the replay: the
        // producer's chain (table, per-wave list walk, histogram, candidate pass, exact recompute, flush) is no shorter than
        // eight per-list tables, and its bound is looser by the margin.  GAMMA_HIP_PROD_CF=1 turns it on (parity-tested).
        static const bool prod_cf_on = getenv("GAMMA_HIP_PROD_CF") != nullptr;
        sb.prod_cf = (cf_ok && !shard && prod_cf_on) ? 1 : 0;
        // the consumers' filter pass on a byte image of the query's table (scan.hip, "byte table")
        const bool c8_on = c8_shape;
        const int c8_mode = 1;
        static const int scan_batch = getenv("GAMMA_HIP_SCAN_BATCH") ? atoi(getenv("GAMMA_HIP_SCAN_BATCH")) : 0;
        sb.batch = scan_batch;
        sb.dbg_part = h->scan_dbg_now;
        sb.c8 = (cf_ok && !sb.prod_cf && c8_on && (cf_span > 0 ? cf_span : P - G) <= 64) ? c8_mode : 0;
        sb.prod_c8 = (one_wg && sb.c8) ? 1 : 0;
        if (one_wg && !sb.c8) return fail(h, GAMMA_HIP_EINVAL, "one workgroup per query without the byte-image pass");
        const bool prod_approx = sb.prod_cf || sb.prod_c8;   // group 0's slab segment does not hold the reference's values
        // two-phase shard search: the producers' bounds out, the reduced (global) bounds back into the ready words
        auto exchange = [&]() -> int {
            gh::launch_bound_export(s, l2, sb.ready, nq, bx->d_bound);
            GH_TRY(bx->reduce(h, nq, l2));
            gh::launch_bound_import(s, l2, bx->d_bound, nq, sb.ready);
            return GAMMA_HIP_OK;
        };
        if (!q8_ok && two) {
            {   // phase 1: the producers alone (one group per query; the plain bounded kernel)
                StageScope t(h, GAMMA_HIP_STAGE_SCAN, true);
                gh::launch_ivfpq_scan_pair(s, l2, d_x, nq, d, M, P, h->w_probe.as<int>(), dis0, h->d_cc, h->scan_st2(l2), h->d_T2,
                                           h->d_list_off, h->d_list_len, h->d_list_mask, nlist, h->d_codes, h->d_ids,
                                           h->w_pair_off.as<int>(), q_stride, h->w_dist.as<float>(), fc.d_tab, fc.d_qf, need_ids, qperm, G, 0,
                                           1, 1, &sb, fuse_ip ? h->d_pqc : nullptr);
            }
            GH_TRY(exchange());
            if (PGM > 1) {   // phase 2: the consumers alone, against the global bounds
                sb.part = 2;
                scan(G, 0, PGM, &sb, false);
                sb.part = 0;
            }
        } else if (!q8_ok) {
            scan(G, 0, PGM, &sb, true);
        } else {
            // producers (the first G probes: exact, they publish the bounds), then the other probes list-major over byte
            // tables -- all of it one "scan launch" for the stage clock and the roofline figure
            StageScope t(h, GAMMA_HIP_STAGE_SCAN, true);
            gh::launch_ivfpq_scan_pair(s, l2, d_x, nq, d, M, P, h->w_probe.as<int>(), dis0, h->d_cc, h->scan_st2(l2), h->d_T2,
                                       h->d_list_off, h->d_list_len, h->d_list_mask, nlist, h->d_codes, h->d_ids,
                                       h->w_pair_off.as<int>(), q_stride, h->w_dist.as<float>(), fc.d_tab, fc.d_qf, need_ids, qperm, G, 0,
                                       1, 0, &sb, q8_fused ? h->d_pqc : nullptr);
            if (two) GH_TRY(exchange());   // (the list-major pass below reads the ready words: now the global bounds)
            GH_CHECK(h, h->w_q8.ensure((size_t)nq * M * 256));
            GH_CHECK(h, h->w_q8meta.ensure((size_t)nq * 4 * sizeof(float)));
            GH_CHECK(h, h->w_q8cand.ensure((size_t)nq * gh::q8_cand_cap(nq) * sizeof(uint32_t)));
            GH_CHECK(h, h->w_q8int.ensure(gh::q8_int_words(nq, P, G, nlist) * sizeof(int)));
            gh::Q8Args qa;
            qa.nq = nq; qa.P = P; qa.G = G; qa.M = M; qa.nlist = nlist;
            qa.mean_len = mean_len;
            qa.probe_list = h->w_probe.as<int>();
            qa.coarse_dis = dis0;
            qa.st2 = h->w_st2.as<float>();
            if (q8_fused) qa.fx = d_x;
            static const bool no_demand = getenv("GAMMA_HIP_NO_Q8_DEMAND") != nullptr;
            if (shard && !no_demand) qa.xd = d_x;
            qa.pqc = h->d_pqc;
            qa.d = d;
            qa.T2 = h->d_T2;
            qa.t2max = h->d_t2max;
            qa.sums = h->d_sums;
            qa.codes = h->d_codes;
            qa.ids = h->d_ids;
            qa.list_off = h->d_list_off;
            qa.list_len = h->d_list_len;
            qa.list_mask = h->d_list_mask;
            qa.pair_off = h->w_pair_off.as<int>();
            qa.ready = sb.ready;
            qa.ftab = fc.d_tab;
            qa.need_ids = need_ids;
            qa.surv = sb.surv;
            qa.gcnt = sb.gcnt;
            qa.cnt_stride = nsl;
            qa.slice_cap = cap;
            qa.rq_list = sb.rq_list;
            qa.rq_count = sb.rq_count;
            qa.q8 = h->w_q8.as<uint8_t>();
            qa.meta = reinterpret_cast<float4*>(h->w_q8meta.p);
            qa.cand = h->w_q8cand.as<uint32_t>();
            qa.iwork = h->w_q8int.as<int>();
            gh::launch_q8_consumers(s, qa);
        }
        // GAMMA_HIP_BOUND_DBG=1: the bounded scan's statistics of the 9th .. 14th call; =shard: of list-shard calls only
        static const bool dbg_any = getenv("GAMMA_HIP_BOUND_DBG") != nullptr;
        static const bool dbg_shard = dbg_any && getenv("GAMMA_HIP_BOUND_DBG")[0] == 's';
        const bool dbg = dbg_any && (!dbg_shard || shard);
        static const int dbg_from = 8;
        static int shown = 0;
        StageScope t(h, GAMMA_HIP_STAGE_SELECT);
        gh::launch_select_final(s, l2, sb.surv, sb.gcnt, nsl, cap, sb.ready, h->w_pair_off.as<int>(), P, nq, R,
                                h->w_pair_base.as<int64_t>(), h->d_ids,
                                h->w_sflag.as<uint8_t>(), out_dis,
                                h->w_cand_pos.as<int>(), out_ids, h->tie.on ? h->w_tcut.as<uint8_t>() : nullptr,
                                h->d_tie_stats, sb.rq_list, sb.rq_count, h->d_bound_stat + 2 * (h->bound_epoch & 1));
        // the counts as of this call, for a later call's decision (16 bytes; ~5 us on the call's critical path, so: the
        // first calls of a kind, then every 16th)
        if ((h->bound_calls < 4 || (h->bound_calls & 15) == 0) && !h->bound_copy_pending) {
            GH_CHECK(h, hipMemcpyAsync(h->pin_bound_stat + 2 * (h->bound_epoch & 1), h->d_bound_stat + 2 * (h->bound_epoch & 1),
                                       2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
            if (!h->bound_copy_ev) GH_CHECK(h, hipEventCreateWithFlags(&h->bound_copy_ev, hipEventDisableTiming));
            GH_CHECK(h, hipEventRecord(h->bound_copy_ev, s));
            h->bound_copy_pending = true;
        }
        h->bound_calls++;
        if (PGN > 1 && !sb.store_all) {
            // queries the slices could not answer: their consumer groups are scored again, distances stored
            // (prod_cf: group 0 too -- its slab segment holds approximate values -- and for the queries without a bound as well)
            StageScope t2(h, GAMMA_HIP_STAGE_SCAN, false);
            if (sb.prod_cf) gh::launch_rq_nobound(s, sb.ready, nq, sb.rq_list, sb.rq_count);
            if (q8_fused) gh::launch_pq_ip_table_rows(s, d_x, d, M, h->d_pqc, h->w_st2.as<float>(), sb.rq_list, sb.rq_count);
            gh::launch_ivfpq_scan_pair(s, l2, d_x, nq, d, M, P, h->w_probe.as<int>(), dis0, h->d_cc,
                                       h->scan_st2(l2), h->d_T2, h->d_list_off, h->d_list_len, h->d_list_mask,
                                       nlist, h->d_codes, h->d_ids, h->w_pair_off.as<int>(), q_stride,
                                       h->w_dist.as<float>(), fc.d_tab, fc.d_qf, need_ids, nullptr, G, prod_approx ? 0 : 1,
                                       prod_approx ? PGN : PGN - 1, shard ? 1 : 0, nullptr, nullptr, sb.rq_list, sb.rq_count);
        }
        h->tie.prod_cf = prod_approx;
        h->tie.need_ids = need_ids;
        h->tie.d_ftab = fc.d_tab;
        h->tie.d_qf = fc.d_qf;
        h->tie.dis0 = dis0;
        gh::launch_select_topk(s, l2, h->w_dist.as<float>(), q_stride, h->w_qtotal.as<int>(), 0,
                               (int)std::min<int64_t>(q_stride, 1 << 30), nq, R,
                               out_dis, h->w_cand_pos.as<int>(), h->w_sflag.as<uint8_t>());
        if (h->tie.on) {
            gh::launch_flag_cut_ties(s, h->w_dist.as<float>(), q_stride, h->w_qtotal.as<int>(), nq, R, out_dis,
                                     h->w_cand_pos.as<int>(), h->w_sflag.as<uint8_t>(), h->w_tcut.as<uint8_t>());
            h->tie.bounded = true;
            h->tie.nsl = nsl;
            h->tie.cap = cap;
            h->tie.slice0_all = sb.prod_c8 != 0;
        }
        if (dbg && shown++ >= dbg_from && shown <= dbg_from + 5) {
            std::vector<uint8_t> hf(nq);
            std::vector<int> hc((size_t)nq * nsl);
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(hf.data(), h->w_sflag.p, nq, hipMemcpyDeviceToHost);
            (void)hipMemcpy(hc.data(), sb.gcnt, hc.size() * sizeof(int), hipMemcpyDeviceToHost);
            std::vector<unsigned long long> hr(nq);
            (void)hipMemcpy(hr.data(), sb.ready, (size_t)nq * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            int64_t nf = 0, tot = 0, mx = 0, nobound = 0;
            for (int i = 0; i < nq; i++) {
                nf += hf[i];
                nobound += (hr[i] >> 32) != 1ull;
            }
            for (size_t i = 0; i < hc.size(); i++) {
                tot += hc[i];
                mx = std::max<int64_t>(mx, hc[i]);
            }
            fprintf(stderr, "scan bound: %lld of %d queries unfiltered (%lld without a bound), survivors per query mean %.1f, "
                    "per slice max %lld; G %d, %d groups, %d slices, q_stride %lld; consumers that gave up waiting so far %llu\n",
                    (long long)nf, nq, (long long)nobound, (double)tot / nq, (long long)mx, G, PGN, nsl, (long long)q_stride,
                    [&] { unsigned long long t = 0; (void)hipMemcpy(&t, h->d_bound_stat + 4, sizeof(t), hipMemcpyDeviceToHost); return t; }());
        }
        gh::launch_map_candidates(s, h->w_cand_pos.as<int>(), nq, R, P, h->w_probe.as<int>(),
                                  h->w_pair_off.as<int>(), h->d_list_off, h->d_ids,
                                  out_ids, h->w_sflag.as<uint8_t>());
    }
    h->tie.G = G;
    h->tie.q_stride = q_stride;
    GH_CHECK(h, hipEventRecord(h->rd_ev[ver], s));
    h->rd_set[ver] = true;
    GH_CHECK(h, hipGetLastError());
    return GAMMA_HIP_OK;
}

// ---- stage B: compute_dis (gamma_index_ivfpq.cc:642-697) ------------------------------
int ivfpq_stage_b(H* h, const gamma_hip_search_params* p, int nq, const float* d_x, int R, int k,
                  const float* cand_dis, const int64_t* cand_ids, float* d_distances,
                  int64_t* d_labels, const int* qperm = nullptr, int tie_mode = 0) {   // 1: flag + replay; 2: flag only
                                                                                         // (merge of shards: w_tcut / w_tlist set up by the caller)
    const bool l2 = p->metric == GAMMA_HIP_METRIC_L2;
    const float neutral = l2 ? 3.402823466e+38f : -3.402823466e+38f;
    hipStream_t s = h->stream;
    StageScope t(h, GAMMA_HIP_STAGE_RERANK);
    // exact ties: the final-stage kernel lists the queries with a tie among their first k+1 distances (or with a
    // tied top-R cut, stage A) and k_tie_replay redoes those the way the reference's heaps do (ties.hip)
    const bool ties = (tie_mode == 1 && h->tie.on) || tie_mode == 2;
    gh::TieFlags tf;
    if (ties) {
        tf.cut = h->w_tcut.as<uint8_t>();
        tf.count = h->w_tlist.as<int>();
        tf.list = h->w_tlist.as<int>() + 1;
        tf.stats = h->d_tie_stats;
    }
    auto replay = [&]() {
        gh::TieReplayArgs a;
        a.list = tf.list;
        a.count = tf.count;
        a.nq = nq;
        a.slab = h->w_dist.as<float>();
        a.q_stride = h->tie.q_stride;
        a.pair_off = h->w_pair_off.as<int>();
        a.pair_base = h->w_pair_base.as<int64_t>();
        a.ids = h->d_ids;
        a.P = p->nprobe;
        a.G = h->tie.G;
        const size_t rq_bytes = (((size_t)nq + 1) * sizeof(int) + 7) & ~(size_t)7;
        unsigned long long* ready = reinterpret_cast<unsigned long long*>(h->w_scnt.as<char>() + rq_bytes);
        a.ready = h->tie.bounded ? ready : nullptr;
        a.surv = h->w_surv.as<unsigned long long>();
        a.gcnt = reinterpret_cast<int*>(ready + nq);
        a.nsl = h->tie.nsl;
        a.slice_cap = h->tie.cap;
        a.slice0_all = h->tie.slice0_all ? 1 : 0;
        a.x = d_x;
        a.d = h->d;
        a.raw = h->d_raw;
        a.nraw = h->nraw;
        a.R = R;
        a.k = k;
        a.has_rank = p->has_rank ? 1 : 0;
        a.min_score = p->min_score;
        a.max_score = p->max_score;
        a.neutral = neutral;
        a.cand_dis = const_cast<float*>(cand_dis);
        a.cand_ids = const_cast<int64_t*>(cand_ids);
        a.distances = d_distances;
        a.labels = d_labels;
        if (h->tie.prod_cf) {
            // the flagged queries' first probe group: the slab holds the producer's approximate values (ScanBound::prod_cf), the
            // replay walks the reference's -- re-scored here, on the search stream, in front of the replay (the tables and the
            // assignment of this call are still in place)
            gh::launch_ivfpq_scan_pair(s, l2, d_x, nq, h->d, h->M, p->nprobe, h->w_probe.as<int>(), h->tie.dis0, h->d_cc,
                                       h->scan_st2(l2), h->d_T2, h->d_list_off, h->d_list_len, h->d_list_mask, h->nlist,
                                       h->d_codes, h->d_ids, h->w_pair_off.as<int>(), h->tie.q_stride, h->w_dist.as<float>(),
                                       static_cast<const gh::FilterDesc*>(h->tie.d_ftab), h->tie.d_qf, h->tie.need_ids, nullptr,
                                       h->tie.G, 0, 1, 0, nullptr, nullptr, tf.list, tf.count);
        }
        static const bool no_side = getenv("GAMMA_HIP_NO_SIDE_STREAM") != nullptr;
        if (h->defer_now && h->side2 && !no_side) {
            // beside whatever the search stream does next that does not touch the replay's inputs (replay_join)
            (void)hipEventRecord(h->ev_rfork, s);
            (void)hipStreamWaitEvent(h->side2, h->ev_rfork, 0);
            gh::launch_tie_replay(h->side2, l2, a);
            (void)hipEventRecord(h->ev_rdone, h->side2);
            h->replay_pending = true;
            h->replay_is_flat = false;
        } else {
            gh::launch_tie_replay(s, l2, a);
        }
    };
    if (p->has_rank) {
        if (!h->d_raw || h->raw_d != h->d) return fail(h, GAMMA_HIP_EINVAL, "has_rank needs the raw store");
        if (h->raw_sparse) return fail(h, GAMMA_HIP_EUNSUPPORTED, "has_rank on a handle that holds its shard's raw rows only: the exact distances travel with the candidates (gamma_hip_ivfpq_shard_exact / _merge_rerank_exact)");
        if (R <= 1024 && (nq >= 256 || ties)) {
            // one fused kernel: exact distances + top-k + output
            gh::launch_rerank_topk(s, l2, d_x, nq, h->d, h->d_raw, h->nraw, cand_ids, R, k, p->min_score,
                                   p->max_score, neutral, d_distances, d_labels, qperm, ties ? &tf : nullptr);
            if (ties && tie_mode == 1) replay();
            GH_CHECK(h, hipGetLastError());
            return GAMMA_HIP_OK;
        }
        GH_CHECK(h, h->w_exact.ensure((size_t)nq * R * sizeof(float)));
        GH_CHECK(h, h->w_selv.ensure((size_t)nq * k * sizeof(float)));
        GH_CHECK(h, h->w_selp.ensure((size_t)nq * k * sizeof(int)));
        gh::launch_rerank_dist(s, l2, d_x, nq, h->d, h->d_raw, h->nraw, cand_ids, R, p->min_score,
                               p->max_score, h->w_exact.as<float>());
        gh::launch_select_topk(s, l2, h->w_exact.as<float>(), R, nullptr, R, R, nq, k,
                               h->w_selv.as<float>(), h->w_selp.as<int>());
        gh::launch_finalize_topk(s, h->w_selv.as<float>(), h->w_selp.as<int>(), nq, k, cand_ids, R, 0,
                                 neutral, d_distances, d_labels);
        if (ties) {
            // recall_num beyond 1024: the flags of the fused kernel, made here -- equal exact distances among the k selected
            // or at the k cut (the row of exact distances against the selection), or a tied top-R cut (stage A)
            GH_CHECK(h, h->w_textra.ensure((size_t)nq));
            GH_CHECK(h, hipMemsetAsync(h->w_textra.p, 0, (size_t)nq, s));
            gh::launch_flag_cut_ties(s, h->w_exact.as<float>(), R, nullptr, nq, k, h->w_selv.as<float>(), h->w_selp.as<int>(),
                                     nullptr, h->w_textra.as<uint8_t>(), R, 1);
            gh::launch_tie_list(s, tf.cut, h->w_textra.as<uint8_t>(), nq, tf.list, tf.count, tf.stats);
            if (tie_mode == 1) replay();
        }
    } else {
        gh::launch_finalize_norank(s, cand_dis, cand_ids, nq, R, k, p->min_score, p->max_score, neutral,
                                   d_distances, d_labels, ties ? &tf : nullptr);
        if (ties && tie_mode == 1) replay();
    }
    GH_CHECK(h, hipGetLastError());
    return GAMMA_HIP_OK;
}

// ---- small batches (nq <= 512; measured cross-over with the regular chain ~1000): four or five launches instead of eleven ---------------------------------
// exact coarse distances + query tables | top-nprobe + slab offsets | scan | top-recall_num + ids + re-rank + top-k
// (kernels.hip k_small_coarse_ip, select.hip k_small_coarse_select / k_small_tail).  Each launch of the regular
// chain costs ~4 us of launch + drain at this size, whatever it computes.
bool ivfpq_small_ok(H* h, const gamma_hip_search_params* p, const FiltCtx& fc, int nq, int R) {
    static const bool off = getenv("GAMMA_HIP_NO_SMALL_PATH") != nullptr;
    static const int max_nq = getenv("GAMMA_HIP_SMALL_MAX") ? atoi(getenv("GAMMA_HIP_SMALL_MAX")) : 512;
    // exact coarse distances (faiss below 20 queries) come from the fused first kernel, which covers 16 queries; the
    // GEMM form (20 queries and more) from the regular matrix kernel
    return !off && h->small_path && nq >= 1 && nq <= max_nq &&
           p->nprobe <= 128 && R <= 1024 && !h->profile && !fc.d_qf && !h->d_list_mask &&
           h->nlist <= 16384 &&
           (int64_t)p->nprobe * std::max(1, h->max_list_len) <= (1 << 22) &&
           // long lists: beyond ~5e7 codes per call the regular chain's bound filter wins (full-size C4, 390 k codes per
           // query: 64 queries 0.46 ms against 1.04, 256 queries 1.63 against 1.33)
           (int64_t)nq * p->nprobe * (h->ntotal / std::max(1, h->nlist)) <= 48000000LL && (!p->has_rank || (h->d_raw && h->raw_d == h->d && !h->raw_sparse));
}

int ivfpq_small(H* h, const gamma_hip_search_params* p, const FiltCtx& fc, int nq, const float* d_x, int R, int k,
                float* d_distances, int64_t* d_labels) {
    const int P = p->nprobe, d = h->d, M = h->M, nlist = h->nlist;
    hipStream_t s = h->stream;
    const int ver = h->cur_ver;
    GH_CHECK(h, hipStreamWaitEvent(s, h->ver_ev[ver], 0));
    GH_CHECK(h, h->w_mat.ensure((size_t)nq * nlist * sizeof(float)));
    GH_CHECK(h, h->w_st2.ensure((size_t)nq * M * 256 * sizeof(float)));
    GH_CHECK(h, h->w_coarse_dis.ensure((size_t)nq * P * sizeof(float)));
    GH_CHECK(h, h->w_probe.ensure((size_t)nq * P * sizeof(int)));
    GH_CHECK(h, h->w_pair_off.ensure((size_t)nq * (P + 1) * sizeof(int)));
    GH_CHECK(h, h->w_pair_base.ensure((size_t)nq * P * sizeof(int64_t)));
    GH_CHECK(h, h->w_qtotal.ensure((size_t)nq * sizeof(int)));
    GH_CHECK(h, h->w_cand_pos.ensure((size_t)nq * R * sizeof(int)));
    GH_CHECK(h, h->w_cand_dis.ensure((size_t)nq * R * sizeof(float)));
    GH_CHECK(h, h->w_cand_ids.ensure((size_t)nq * R * sizeof(int64_t)));
    const int64_t q_stride = (std::max<int64_t>(1, (int64_t)P * std::max(1, h->max_list_len)) + 3) & ~(int64_t)3;
    GH_CHECK(h, h->w_dist.ensure((size_t)nq * q_stride * sizeof(float)));
    // (folding the selection into the first launch -- last workgroup done selects -- was tried: the device-scope
    // release / acquire it needs costs more than the launch it saves, 24 us against 4 + 8: the XCDs' L2s are
    // written back and invalidated either way)
    // long lists: the scan walks a work list of (query, probe, chunk of the list) units written by the selection kernel,
    // pieces of even size for a grid that fills the chip, instead of one workgroup per pair that runs for as long as its
    // list is (small_presel: tests force the path on short lists)
    static const int chunk_env = getenv("GAMMA_HIP_SMALL_CHUNK") ? atoi(getenv("GAMMA_HIP_SMALL_CHUNK")) : 512;
    int chunk_len = 0, max_units = 0;
    uint32_t* d_units = nullptr;
    int* d_nunits = nullptr;
    if (h->ntotal / std::max(1, nlist) > 1024 || h->max_list_len > 8192 || h->small_presel > 0) {
        chunk_len = h->small_presel > 0 ? 512 : std::max(512, (chunk_env + 511) & ~511);
        const int64_t mu = (int64_t)nq * P * (1 + (int64_t)h->max_list_len / chunk_len);
        max_units = (int)std::min<int64_t>(mu, INT32_MAX);
        GH_CHECK(h, h->w_lm_units.ensure((size_t)mu * sizeof(uint32_t)));
        GH_CHECK(h, h->w_lm_cnt.ensure(64));
        d_units = h->w_lm_units.as<uint32_t>();
        d_nunits = h->w_lm_cnt.as<int>();
    }
    if (p->coarse_mode == 1) {
        if (d_nunits) GH_CHECK(h, hipMemsetAsync(d_nunits, 0, sizeof(int), s));
        gh::launch_l2_gemmform(s, d_x, nq, d, h->d_cc, nlist, nullptr, h->d_cc_norms, h->w_mat.as<float>(), nlist, true);
        gh::launch_pq_ip_table(s, d_x, nq, d, M, h->d_pqc, h->w_st2.as<float>());
    } else if (!gh::launch_small_coarse_ip(s, d_x, nq, d, h->d_cc, nlist, h->w_mat.as<float>(), M, h->d_pqc,
                                           h->w_st2.as<float>(), d_nunits)) {
        // exact coarse distances for more than 16 queries (an explicit coarse_mode 0, or a combined batch of requests
        // that are each below faiss's 20-query switch): the regular chain's kernel, then the small chain
        if (d_nunits) GH_CHECK(h, hipMemsetAsync(d_nunits, 0, sizeof(int), s));
        gh::launch_pairwise(s, true, d_x, nq, d, h->d_cc, nlist, h->w_mat.as<float>(), nlist);
        gh::launch_pq_ip_table(s, d_x, nq, d, M, h->d_pqc, h->w_st2.as<float>());
    }
    const bool l2 = p->metric == GAMMA_HIP_METRIC_L2;
    if (!l2) GH_CHECK(h, h->w_pair_ip.ensure((size_t)nq * P * sizeof(float)));
    gh::launch_small_coarse_select(s, h->w_mat.as<float>(), nlist, nq, P, h->w_coarse_dis.as<float>(), h->w_probe.as<int>(),
                                   h->d_list_len, h->d_list_mask, h->d_list_off, h->w_pair_off.as<int>(),
                                   h->w_qtotal.as<int>(), h->w_pair_base.as<int64_t>(), d_x, h->d_cc, d,
                                   l2 ? nullptr : h->w_pair_ip.as<float>(), d_units, d_nunits, chunk_len,
                                   tie_on(p) ? 1 : 0, h->d_tie_stats);
    h->scan_pairs += (int64_t)nq * P;
    const int need_ids = (!h->prefiltered && (fc.any_clause || (h->d_bitmap && h->bitmap_any) || h->n_moved > 0)) ? 1 : 0;
    gh::launch_ivfpq_scan_pair(s, l2, d_x, nq, d, M, P, h->w_probe.as<int>(),
                               l2 ? h->w_coarse_dis.as<float>() : h->w_pair_ip.as<float>(), h->d_cc,
                               h->scan_st2(l2), h->d_T2, h->d_list_off, h->d_list_len, h->d_list_mask, nlist,
                               h->d_codes, h->d_ids, h->w_pair_off.as<int>(), q_stride, h->w_dist.as<float>(), fc.d_tab,
                               fc.d_qf, need_ids, nullptr, 1, 0, P, 0, nullptr, nullptr, reinterpret_cast<const int*>(d_units), d_nunits,
                               chunk_len, max_units);
    const float neutral = l2 ? 3.402823466e+38f : -3.402823466e+38f;
    // long candidate rows (expected nprobe x 1.5 mean list lengths beyond what one workgroup keeps in registers): a first
    // selection over slices of the row by several workgroups per query, then the tail among their survivors
    int smax = 0;
    {
        const int64_t slice = 16384;   // select.hip SM_SLICE
        const int64_t est = (int64_t)P * (h->ntotal / std::max(1, nlist)) * 3 / 2;
        const int64_t bound = (int64_t)P * std::max(1, h->max_list_len);
        if (h->small_presel > 0) smax = h->small_presel;
        else if (est > slice) smax = (int)std::min<int64_t>(std::min<int64_t>(64, (bound + slice - 1) / slice), std::max(2, 4096 / nq));
        if (smax > 0) {
            GH_CHECK(h, h->w_selv.ensure((size_t)nq * smax * R * sizeof(float)));
            GH_CHECK(h, h->w_selp.ensure(((size_t)nq * smax * R + (size_t)nq * smax) * sizeof(int)));   // + a cut flag per slice
        }
    }
    // exact ties: a query with a tie at the recall_num or k cut is replayed through the reference's heaps inside the
    // tail kernel (tie_dev.h), from the whole slab row
    gh::TieReplayArgs tr;
    const bool ties = tie_on(p) && R <= gh::tie_small_max_k();   // (ivfpq_small_ok: recall_num <= 1024)
    if (ties) {
        tr.list = nullptr;
        tr.count = nullptr;
        tr.nq = nq;
        tr.slab = h->w_dist.as<float>();
        tr.q_stride = q_stride;
        tr.pair_off = h->w_pair_off.as<int>();
        tr.pair_base = h->w_pair_base.as<int64_t>();
        tr.ids = h->d_ids;
        tr.P = P;
        tr.G = 1;
        tr.ready = nullptr;
        tr.surv = nullptr;
        tr.gcnt = nullptr;
        tr.nsl = 0;
        tr.slice_cap = 0;
        tr.x = d_x;
        tr.d = d;
        tr.raw = h->d_raw;
        tr.nraw = h->nraw;
        tr.R = R;
        tr.k = k;
        tr.has_rank = p->has_rank ? 1 : 0;
        tr.min_score = p->min_score;
        tr.max_score = p->max_score;
        tr.neutral = neutral;
        tr.cand_dis = h->w_cand_dis.as<float>();
        tr.cand_ids = h->w_cand_ids.as<int64_t>();
        tr.distances = d_distances;
        tr.labels = d_labels;
    }
    gh::launch_small_tail(s, l2, h->w_dist.as<float>(), q_stride, h->w_qtotal.as<int>(), nq, R, P, h->w_probe.as<int>(),
                          h->w_pair_off.as<int>(), h->d_list_off, h->d_ids, h->w_cand_dis.as<float>(),
                          h->w_cand_pos.as<int>(), h->w_cand_ids.as<int64_t>(), p->has_rank ? 1 : 0, d_x, d, h->d_raw,
                          h->nraw, k, p->min_score, p->max_score, neutral, d_distances, d_labels, smax,
                          smax ? h->w_selv.as<float>() : nullptr, smax ? h->w_selp.as<int>() : nullptr, 0,
                          ties ? &tr : nullptr, h->d_tie_stats);
    h->tie = H::TieCtx();
    h->tie.G = 1;
    h->tie.q_stride = q_stride;
    h->last_qperm = nullptr;
    GH_CHECK(h, hipEventRecord(h->rd_ev[ver], s));
    h->rd_set[ver] = true;
    GH_CHECK(h, hipGetLastError());
    return GAMMA_HIP_OK;
}

int ivfpq_check(H* h, const gamma_hip_search_params* p, int nq, int k) {
    GH_TRY(check_params(h, p, nq, k));
    if (!h->ivf_init || h->ivfflat) return fail(h, GAMMA_HIP_EINVAL, "ivfpq not initialised");
    if (!h->trained) return fail(h, GAMMA_HIP_ENOTTRAINED, "ivfpq not trained");
    if (p->nprobe <= 0 || p->nprobe > h->nlist) return fail(h, GAMMA_HIP_EINVAL, "nprobe out of range");
    if (std::max(p->recall_num, k) > 4096) return fail(h, GAMMA_HIP_EINVAL, "recall_num > 4096 unsupported");
    return GAMMA_HIP_OK;
}

// queries per internal chunk: the coarse distance matrix (nlist floats per query) and the ADC distance
// slab (nprobe x longest list floats per query) each stay inside the workspace budget
int coarse_chunk(H* h, int nq) {
    const int64_t by_mat = (int64_t)(h->dist_budget_bytes / ((size_t)h->nlist * sizeof(float)));
    return (int)std::max<int64_t>(1, std::min<int64_t>(by_mat, nq));
}
int scan_chunk(H* h, int nq, int P) {
    const int64_t q_stride = std::max<int64_t>(1, (int64_t)P * std::max(1, h->max_list_len));
    const int64_t by_dist = (int64_t)(h->dist_budget_bytes / (q_stride * sizeof(float)));
    return (int)std::max<int64_t>(1, std::min<int64_t>(by_dist, nq));
}
int query_chunk(H* h, int nq, int P) { return std::min(coarse_chunk(h, nq), scan_chunk(h, nq, P)); }

// Large filtered batches: the lists are cut down to the entries that pass -- once per call, the predicate does not
// depend on the query (kernels.hip k_compact_lists) -- and the call runs unfiltered over the shadow lists.  Worth it
// when the call tests several times as many entries as the index holds; GAMMA_HIP_LIST_COMPACT=0/1 never / always.
// The handle's list pointers are swapped until the ListCompaction object goes out of scope (the caller holds the
// search lock and h->mu for the whole enqueue).
struct ListCompaction {
    H* h;
    uint8_t* codes;
    int64_t* ids;
    int* len;
    float* sums;
    bool on = false;
    explicit ListCompaction(H* h_) : h(h_), codes(h_->d_codes), ids(h_->d_ids), len(h_->d_list_len), sums(h_->d_sums) {}
    ~ListCompaction() {
        if (on) {
            h->d_codes = codes;
            h->d_ids = ids;
            h->d_list_len = len;
            h->d_sums = sums;
            h->prefiltered = false;
            h->cmp_has_sums = false;
        }
    }
};

int compact_lists_for_call(H* h, FiltCtx* fc, int64_t est, bool allowed, ListCompaction* lc) {
    const bool need = fc->any_clause || (h->d_bitmap && h->bitmap_any) || h->n_moved > 0;
    const char* env = getenv("GAMMA_HIP_LIST_COMPACT");
    const bool want = env ? atoi(env) != 0 : est >= 4 * std::max<int64_t>(1, h->ntotal);
    if (!(need && want && allowed && !fc->d_qf && !h->d_list_mask && h->arena_cap > 0)) return GAMMA_HIP_OK;
    // the shadow arena is an optimisation: without the memory for it the call runs over the lists as they are, testing
    // the predicate per scored code (same results)
    if (h->w_cmp_codes.ensure((size_t)h->arena_cap * h->code_size) != hipSuccess ||
        h->w_cmp_ids.ensure((size_t)h->arena_cap * sizeof(int64_t)) != hipSuccess ||
        h->w_cmp_len.ensure((size_t)h->nlist * sizeof(int)) != hipSuccess) {
        (void)hipGetLastError();
        h->w_cmp_codes.release();
        h->w_cmp_ids.release();
        return GAMMA_HIP_OK;
    }
    // the per-code sums of the scan's filter pass go along (same offsets), so the pass stays on over the shadow lists
    const bool with_sums = h->d_sums && h->w_cmp_sums.ensure((size_t)h->arena_cap * sizeof(float)) == hipSuccess;
    if (!with_sums) (void)hipGetLastError();
    // standing deletes (no clause of the call's own): the shadow lists of the last call are still right unless a writer
    // has run since -- Add / Update / Delete / bitmap / compaction all count in write_gen
    static const bool no_cache = getenv("GAMMA_HIP_NO_COMPACT_CACHE") != nullptr;
    const bool reuse = !no_cache && !fc->any_clause && h->cmp_gen == h->write_gen && h->cmp_sums_built == with_sums;
    GH_CHECK(h, hipStreamWaitEvent(h->stream, h->ver_ev[h->cur_ver], 0));   // the version's lists are in place
    if (!reuse) {
        GH_TRY(replay_join(h));   // a deferred replay may still read the shadow lists of the previous call
        StageScope t(h, GAMMA_HIP_STAGE_SCAN, false);   // profiled as part of the scan it shortens
        gh::launch_compact_lists(h->stream, h->d_list_off, h->d_list_len, h->nlist, h->d_codes, h->d_ids, h->code_size,
                                 fc->d_tab, h->w_cmp_codes.as<uint8_t>(), h->w_cmp_ids.as<int64_t>(), h->w_cmp_len.as<int>(),
                                 with_sums ? h->d_sums : nullptr, with_sums ? h->w_cmp_sums.as<float>() : nullptr);
        h->cmp_gen = fc->any_clause ? 0 : h->write_gen;   // shadow lists under a request's own clauses serve that request only
        h->cmp_sums_built = with_sums;
    }
    if (with_sums) h->d_sums = h->w_cmp_sums.as<float>();
    h->cmp_has_sums = with_sums;
    h->cmp_by_clause = fc->any_clause;
    h->d_codes = h->w_cmp_codes.as<uint8_t>();
    h->d_ids = h->w_cmp_ids.as<int64_t>();
    h->d_list_len = h->w_cmp_len.as<int>();
    h->prefiltered = true;
    lc->on = true;
    fc->any_clause = false;
    return GAMMA_HIP_OK;
}

// given != nullptr: the filter context of a combined batch (p's own filter clauses are ignored)
int ivfpq_search_device_locked(H* h, const gamma_hip_search_params* p, int nq, const float* d_x, int k,
                               float* d_distances, int64_t* d_labels, const FiltCtx* given) {
    GH_TRY(ivfpq_check(h, p, nq, k));
    gamma_hip_search_params pp;
    GH_TRY(resolve_ties(h, p, &pp, p->nprobe <= gh::tie_replay_max_probes(), "exact_ties = 1 with nprobe > 1024"));
    p = &pp;
    if (k <= 0 || nq == 0) {   // gamma_index_ivfpq.cc:753-756
        // (the deferred-replay contract: the previous call is complete after ANY next search call, an empty one too)
        if (h->replay_pending) {
            GH_CHECK(h, hipSetDevice(h->device));
            GH_TRY(replay_join(h));
        }
        return GAMMA_HIP_OK;
    }
    GH_CHECK(h, hipSetDevice(h->device));
    const int R = std::max(p->recall_num, k);
    FiltCtx fc;
    if (given) {
        fc = *given;
    } else {
        gh::FilterDesc filt;
        GH_TRY(build_filter(h, p, &filt, nullptr, (int64_t)nq * p->nprobe * (h->ntotal / std::max(1, h->nlist))));
        GH_TRY(filt_ctx_single(h, filt, &fc));
    }
    // faiss picks the coarse path from the size of the WHOLE call (faiss:utils/distances.cpp:346);
    // the internal chunks must not re-decide it
    if (pp.coarse_mode < 0) pp.coarse_mode = nq < 20 ? 0 : 1;
    if (pp.coarse_mode == 1 && blas_form_not_restated(nq, h->nlist, h->d)) h->blas_unrestated++;
    if (ivfpq_small_ok(h, p, fc, nq, R)) {
        GH_TRY(replay_join(h));
        GH_TRY(ivfpq_small(h, p, fc, nq, d_x, R, k, d_distances, d_labels));
        h->last_nq = nq;
        h->last_P = p->nprobe;
        h->last_R = R;
        return GAMMA_HIP_OK;
    }
    ListCompaction restore(h);
    GH_TRY(compact_lists_for_call(h, &fc, (int64_t)nq * p->nprobe * (h->ntotal / std::max(1, h->nlist)),
                                  !given, &restore));
    int chunk = scan_chunk(h, nq, p->nprobe);
    const int P = p->nprobe;
    // long lists (C4: 64 probes x lists of tens of thousands) make the ADC slab the limit: the coarse
    // quantizer then still runs over the whole call (one GEMM instead of one per slab chunk)
    const bool coarse_first = chunk < nq;
    struct StrideScope {   // the measured stride holds for this call only
        H* h;
        ~StrideScope() { h->q_stride_cap = 0; }
    } stride_scope{h};
    if (coarse_first) {
        GH_CHECK(h, h->w_full_cdis.ensure((size_t)nq * P * sizeof(float)));
        GH_CHECK(h, h->w_full_probe.ensure((size_t)nq * P * sizeof(int)));
        const int cc = coarse_chunk(h, nq);
        for (int q0 = 0; q0 < nq; q0 += cc)
            GH_TRY(ivfpq_coarse(h, p, std::min(cc, nq - q0), d_x + (size_t)q0 * h->d,
                                h->w_full_cdis.as<float>() + (size_t)q0 * P, h->w_full_probe.as<int>() + (size_t)q0 * P));
        // The general slab stride is nprobe x the LONGEST list; the batch's longest candidate row is what it needs (full-size
        // C4: the longest list is ~5 x the mean, so the budget cut 8192 queries into three chunks -- and the list-major pass
        // of a third of the batch finds 11 queries per list where the whole batch has 32: half-empty tiles of 8).  The
        // assignment is on the device: measure, read one word back (the call is tens of milliseconds long), size the
        // chunks by it -- what the list shards do (gamma_hip_ivfpq_search_shard_preassigned).
        static const bool no_measure = getenv("GAMMA_HIP_NO_ROW_MEASURE") != nullptr;
        if (!no_measure && !given) {
            GH_TRY(coarse_join(h));
            GH_CHECK(h, h->w_shard_cut.ensure(std::max<size_t>((size_t)nq, 16)));
            gh::launch_max_local_total(h->stream, h->w_full_probe.as<int>(), nq, P, h->d_list_len, h->d_list_mask, h->nlist,
                                       h->w_shard_cut.as<int>());
            int mx[2] = {0, 0};
            GH_CHECK(h, hipMemcpyAsync(mx, h->w_shard_cut.p, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
            GH_CHECK(h, hipStreamSynchronize(h->stream));
            h->q_stride_cap = (std::max<int64_t>(mx[0], 1) + 3) & ~(int64_t)3;
            const int64_t by_dist = (int64_t)(h->dist_budget_bytes / ((size_t)h->q_stride_cap * sizeof(float)));
            chunk = (int)std::max<int64_t>(chunk, std::min<int64_t>(by_dist, nq));
        }
    }
    // a call of several chunks: the tie replay of chunk i runs on its own stream beside chunk i + 1 (what
    // gamma_hip_set_deferred_replay does across calls); a caller that has not asked for that gets the join at the end
    struct DeferScope {
        H* h;
        bool saved;
        ~DeferScope() { h->defer_now = saved; }
    } defer_scope{h, h->defer_now};
    const bool defer_inside = chunk < nq && !h->defer_now && h->side2 != nullptr;
    if (defer_inside) h->defer_now = true;
    for (int q0 = 0; q0 < nq; q0 += chunk) {
        const int nc = std::min(chunk, nq - q0);
        if (coarse_first)
            GH_TRY(ivfpq_stage_a(h, p, fc.at(q0), nc, d_x + (size_t)q0 * h->d, R,
                                 h->w_full_cdis.as<float>() + (size_t)q0 * P, h->w_full_probe.as<int>() + (size_t)q0 * P));
        else
            GH_TRY(ivfpq_stage_a(h, p, fc.at(q0), nc, d_x + (size_t)q0 * h->d, R));
        GH_TRY(ivfpq_stage_b(h, p, nc, d_x + (size_t)q0 * h->d, R, k, h->w_cand_dis.as<float>(),
                             h->w_cand_ids.as<int64_t>(), d_distances + (size_t)q0 * k,
                             d_labels + (size_t)q0 * k, getenv("GAMMA_HIP_NO_RERANK_ORDER") ? nullptr : h->last_qperm,
                             /*tie_mode=*/1));
        h->last_nq = nc;
    }
    if (defer_inside) GH_TRY(replay_join(h));
    h->last_P = p->nprobe;
    h->last_R = R;
    return GAMMA_HIP_OK;
}

// ---- IVFFLAT (index/impl/gamma_index_ivfflat.cc:392-567) ------------------------------------------------------
// coarse quantizer (the IVFPQ one) -> slab offsets -> exact distance of every entry of the probed lists
// (k_ivfflat_scan) -> top-k of the slab in (distance, scan position) order -> ids.  The reference's k-heap keeps
// the same k entries (up to its order inside exact ties).
// IVFFLAT, small batches: the chain of ivfpq_small without tables and re-rank -- exact coarse distances | top-nprobe +
// slab offsets | exact distances of the probed lists' rows, one workgroup per pair | top-k + ids + score window
// the replay of a query of the exact scanners (IVFFLAT: P > 0, the slab in probe order; flat: P == 0, fixed_n rows in
// vid order): the k-heap IS the result heap, score window and filters are already in the slab (sentinels)
void flat_tie_args(H* h, gh::TieReplayArgs* tr, bool l2, int nq, int64_t q_stride, int P, int k, const float* d_x, float* cand_dis,
                   int64_t* cand_ids, float* d_distances, int64_t* d_labels, int fixed_n = 0) {
    tr->list = nullptr;
    tr->count = nullptr;
    tr->nq = nq;
    tr->slab = h->w_dist.as<float>();
    tr->q_stride = q_stride;
    tr->pair_off = P > 0 ? h->w_pair_off.as<int>() : nullptr;
    tr->pair_base = P > 0 ? h->w_pair_base.as<int64_t>() : nullptr;
    tr->ids = h->d_ids;
    tr->P = P;
    tr->G = 1;
    tr->ready = nullptr;
    tr->surv = nullptr;
    tr->gcnt = nullptr;
    tr->nsl = 0;
    tr->slice_cap = 0;
    tr->x = d_x;
    tr->d = h->d;
    tr->raw = h->d_raw;
    tr->nraw = h->nraw;
    tr->R = k;
    tr->k = k;
    tr->has_rank = 0;
    tr->min_score = -INFINITY;
    tr->max_score = INFINITY;
    tr->neutral = l2 ? 3.402823466e+38f : -3.402823466e+38f;
    tr->cand_dis = cand_dis;
    tr->cand_ids = cand_ids;
    tr->distances = d_distances;
    tr->labels = d_labels;
    tr->pop_push = 1;
    tr->fixed_n = fixed_n;
}

int ivfflat_small(H* h, const gamma_hip_search_params* p, const FiltCtx& fc, int nq, const float* d_x, int k,
                  float* d_distances, int64_t* d_labels) {
    const int P = p->nprobe, d = h->d, nlist = h->nlist;
    const bool l2 = p->metric == GAMMA_HIP_METRIC_L2;
    hipStream_t s = h->stream;
    const int ver = h->cur_ver;
    GH_CHECK(h, hipStreamWaitEvent(s, h->ver_ev[ver], 0));
    GH_CHECK(h, h->w_mat.ensure((size_t)nq * nlist * sizeof(float)));
    GH_CHECK(h, h->w_coarse_dis.ensure((size_t)nq * P * sizeof(float)));
    GH_CHECK(h, h->w_probe.ensure((size_t)nq * P * sizeof(int)));
    GH_CHECK(h, h->w_pair_off.ensure((size_t)nq * (P + 1) * sizeof(int)));
    GH_CHECK(h, h->w_pair_base.ensure((size_t)nq * P * sizeof(int64_t)));
    GH_CHECK(h, h->w_qtotal.ensure((size_t)nq * sizeof(int)));
    GH_CHECK(h, h->w_cand_pos.ensure((size_t)nq * k * sizeof(int)));
    GH_CHECK(h, h->w_cand_dis.ensure((size_t)nq * k * sizeof(float)));
    GH_CHECK(h, h->w_cand_ids.ensure((size_t)nq * k * sizeof(int64_t)));
    const int64_t q_stride = (std::max<int64_t>(1, (int64_t)P * std::max(1, h->max_list_len)) + 3) & ~(int64_t)3;
    GH_CHECK(h, h->w_dist.ensure((size_t)nq * q_stride * sizeof(float)));
    if (p->coarse_mode == 1) {
        gh::launch_l2_gemmform(s, d_x, nq, d, h->d_cc, nlist, nullptr, h->d_cc_norms, h->w_mat.as<float>(), nlist, true);
    } else if (!gh::launch_small_coarse_ip(s, d_x, nq, d, h->d_cc, nlist, h->w_mat.as<float>(), 0, nullptr, nullptr)) {
        gh::launch_pairwise(s, true, d_x, nq, d, h->d_cc, nlist, h->w_mat.as<float>(), nlist);   // exact, more than 16 queries
    }
    gh::launch_small_coarse_select(s, h->w_mat.as<float>(), nlist, nq, P, h->w_coarse_dis.as<float>(), h->w_probe.as<int>(),
                                   h->d_list_len, h->d_list_mask, h->d_list_off, h->w_pair_off.as<int>(),
                                   h->w_qtotal.as<int>(), h->w_pair_base.as<int64_t>(), nullptr, nullptr, 0, nullptr, nullptr,
                                   nullptr, 0, tie_on(p) ? 1 : 0, h->d_tie_stats);
    const int need_filter = (fc.any_clause || (h->d_bitmap && h->bitmap_any)) ? 1 : 0;
    gh::launch_ivfflat_scan(s, l2, d_x, nq, d, P, h->w_pair_off.as<int>(), h->w_pair_base.as<int64_t>(), h->d_ids, h->d_raw,
                            h->nraw, q_stride, h->w_dist.as<float>(), fc.d_tab, need_filter, p->min_score, p->max_score);
    int smax = 0;   // long candidate rows: two-level selection (ivfpq_small)
    {
        const int64_t slice = 16384;
        const int64_t est = (int64_t)P * (h->ntotal / std::max(1, nlist)) * 3 / 2;
        const int64_t bound = (int64_t)P * std::max(1, h->max_list_len);
        if (h->small_presel > 0) smax = h->small_presel;
        else if (est > slice) smax = (int)std::min<int64_t>(std::min<int64_t>(64, (bound + slice - 1) / slice), std::max(2, 4096 / nq));
        if (smax > 0) {
            GH_CHECK(h, h->w_selv.ensure((size_t)nq * smax * k * sizeof(float)));
            GH_CHECK(h, h->w_selp.ensure(((size_t)nq * smax * k + (size_t)nq * smax) * sizeof(int)));   // + a cut flag per slice
        }
    }
    const float neutral = l2 ? 3.402823466e+38f : -3.402823466e+38f;
    // exact ties: a query with equal distances at the k cut or among its k results is replayed through the scanner's
    // heap (heap_pop + heap_push per accepted entry, heap_reorder: gamma_index_ivfflat.h:52-75) inside the tail kernel
    gh::TieReplayArgs tr;
    const bool ties = tie_on(p) && k <= gh::tie_small_max_k();   // (the caller's gate: k <= 1024)
    if (ties) flat_tie_args(h, &tr, l2, nq, q_stride, P, k, d_x, h->w_cand_dis.as<float>(), h->w_cand_ids.as<int64_t>(), d_distances,
                            d_labels);
    gh::launch_small_tail(s, l2, h->w_dist.as<float>(), q_stride, h->w_qtotal.as<int>(), nq, k, P, h->w_probe.as<int>(),
                          h->w_pair_off.as<int>(), h->d_list_off, h->d_ids, h->w_cand_dis.as<float>(),
                          h->w_cand_pos.as<int>(), h->w_cand_ids.as<int64_t>(), 0, d_x, d, h->d_raw, h->nraw, k, p->min_score,
                          p->max_score, neutral, d_distances, d_labels, smax, smax ? h->w_selv.as<float>() : nullptr,
                          smax ? h->w_selp.as<int>() : nullptr, 0, ties ? &tr : nullptr, h->d_tie_stats);
    GH_CHECK(h, hipEventRecord(h->rd_ev[ver], s));
    h->rd_set[ver] = true;
    h->last_nq = nq;
    h->last_P = P;
    h->last_R = k;
    GH_CHECK(h, hipGetLastError());
    return GAMMA_HIP_OK;
}

int ivfflat_search_device_locked(H* h, const gamma_hip_search_params* p, int nq, const float* d_x, int k,
                                 float* d_distances, int64_t* d_labels) {
    if (h->raw_sparse) return fail(h, GAMMA_HIP_EUNSUPPORTED, "the raw store holds this shard's rows only (gamma_hip_raw_put)");
    GH_TRY(check_params(h, p, nq, k));
    gamma_hip_search_params pp;
    GH_TRY(resolve_ties(h, p, &pp, p->nprobe <= gh::tie_replay_max_probes(), "exact_ties = 1 with nprobe > 1024"));
    p = &pp;
    if (!h->ivf_init || !h->ivfflat) return fail(h, GAMMA_HIP_EINVAL, "ivfflat not initialised");
    if (!h->trained) return fail(h, GAMMA_HIP_ENOTTRAINED, "ivfflat not trained");
    if (p->nprobe <= 0 || p->nprobe > h->nlist) return fail(h, GAMMA_HIP_EINVAL, "nprobe out of range");
    if (!h->d_raw || h->raw_d != h->d) return fail(h, GAMMA_HIP_EINVAL, "ivfflat needs the raw store");
    if (k <= 0 || nq == 0) return GAMMA_HIP_OK;
    GH_CHECK(h, hipSetDevice(h->device));
    const bool l2 = p->metric == GAMMA_HIP_METRIC_L2;
    const float neutral = l2 ? 3.402823466e+38f : -3.402823466e+38f;
    gh::FilterDesc filt;
    GH_TRY(build_filter(h, p, &filt, nullptr, (int64_t)nq * p->nprobe * (h->ntotal / std::max(1, h->nlist))));
    FiltCtx fc;
    GH_TRY(filt_ctx_single(h, filt, &fc));
    if (pp.coarse_mode < 0) pp.coarse_mode = nq < 20 ? 0 : 1;   // faiss:utils/distances.cpp:303,346, whole call
    if (pp.coarse_mode == 1 && blas_form_not_restated(nq, h->nlist, h->d)) h->blas_unrestated++;
    const int P = pp.nprobe, nlist = h->nlist;
    hipStream_t s = h->stream;
    {   // small batches, as long as the pair-per-workgroup scan is the one that would run anyway (below 2 nlist pairs)
        static const bool off = getenv("GAMMA_HIP_NO_SMALL_PATH") != nullptr;
        if (!off && h->small_path && nq <= 512 && P <= 128 && k <= 1024 && !h->profile &&
            !fc.d_qf && !h->d_list_mask && nlist <= 16384 && (int64_t)nq * P < 2 * (int64_t)nlist &&
            (int64_t)P * std::max(1, h->max_list_len) <= (1 << 22))
            return ivfflat_small(h, &pp, fc, nq, d_x, k, d_distances, d_labels);
    }
    ListCompaction restore(h);
    GH_TRY(compact_lists_for_call(h, &fc, (int64_t)nq * P * (h->ntotal / std::max(1, nlist)), true, &restore));
    const int chunk = scan_chunk(h, nq, P);
    const int need_filter = (!h->prefiltered && (fc.any_clause || (h->d_bitmap && h->bitmap_any))) ? 1 : 0;
    for (int q0 = 0; q0 < nq; q0 += chunk) {
        const int nc = std::min(chunk, nq - q0);
        const float* xq = d_x + (size_t)q0 * h->d;
        const int ver = h->cur_ver;
        GH_CHECK(h, hipStreamWaitEvent(s, h->ver_ev[ver], 0));
        GH_CHECK(h, h->w_pair_off.ensure((size_t)nc * (P + 1) * sizeof(int)));
        GH_CHECK(h, h->w_pair_base.ensure((size_t)nc * P * sizeof(int64_t)));
        GH_CHECK(h, h->w_qtotal.ensure((size_t)nc * sizeof(int)));
        GH_CHECK(h, h->w_cand_pos.ensure((size_t)nc * k * sizeof(int)));
        GH_CHECK(h, h->w_cand_dis.ensure((size_t)nc * k * sizeof(float)));
        GH_CHECK(h, h->w_cand_ids.ensure((size_t)nc * k * sizeof(int64_t)));
        GH_TRY(ivfpq_coarse(h, &pp, nc, xq));
        gh::launch_pair_offsets(s, h->w_probe.as<int>(), nc, P, h->d_list_len, h->d_list_mask, nlist,
                                h->w_pair_off.as<int>(), h->w_qtotal.as<int>(), nullptr, h->d_list_off,
                                h->w_pair_base.as<int64_t>());
        const int64_t q_stride = (std::max<int64_t>(1, (int64_t)P * std::max(1, h->max_list_len)) + 3) & ~(int64_t)3;
        GH_CHECK(h, h->w_dist.ensure((size_t)nc * q_stride * sizeof(float)));
        {
            StageScope t(h, GAMMA_HIP_STAGE_SCAN);
            // enough (query, probe) pairs that lists are shared: list-major (ivfflat.hip), a list's rows are read once
            // for all the queries probing it; else one workgroup per pair
            static const bool no_lm = getenv("GAMMA_HIP_NO_IVFFLAT_LM") != nullptr;
            if (!no_lm && gh::ivfflat_lm_supported(h->d) && (int64_t)nc * P >= 2 * (int64_t)nlist && !h->d_list_mask) {
                GH_CHECK(h, h->w_lm_units.ensure(gh::ivfflat_lm_scratch_bytes(nc, P, nlist)));
                gh::launch_ivfflat_lm(s, l2, xq, nc, h->d, P, h->w_probe.as<int>(), h->w_pair_off.as<int>(), h->d_list_off,
                                      h->d_list_len, nlist, h->d_ids, h->d_raw, h->nraw, q_stride, h->w_dist.as<float>(),
                                      fc.d_tab, need_filter, p->min_score, p->max_score, h->w_lm_units.p);
            } else {
                gh::launch_ivfflat_scan(s, l2, xq, nc, h->d, P, h->w_pair_off.as<int>(), h->w_pair_base.as<int64_t>(),
                                        h->d_ids, h->d_raw, h->nraw, q_stride, h->w_dist.as<float>(), fc.d_tab, need_filter,
                                        p->min_score, p->max_score);
            }
        }
        {
            StageScope t(h, GAMMA_HIP_STAGE_SELECT);
            gh::launch_select_topk(s, l2, h->w_dist.as<float>(), q_stride, h->w_qtotal.as<int>(), 0,
                                   (int)std::min<int64_t>(q_stride, 1 << 30), nc, k, h->w_cand_dis.as<float>(),
                                   h->w_cand_pos.as<int>());
            gh::launch_map_candidates(s, h->w_cand_pos.as<int>(), nc, k, P, h->w_probe.as<int>(), h->w_pair_off.as<int>(),
                                      h->d_list_off, h->d_ids, h->w_cand_ids.as<int64_t>());
            const bool ties = tie_on(p);
            gh::TieFlags tf;
            if (ties) {
                // equal distances at the k cut or among the k selected: the scanner's heap decides (ties.hip)
                GH_CHECK(h, h->w_tcut.ensure((size_t)nc));
                GH_CHECK(h, h->w_tlist.ensure(((size_t)nc + 1) * sizeof(int)));
                GH_CHECK(h, hipMemsetAsync(h->w_tlist.p, 0, sizeof(int), s));
                GH_CHECK(h, hipMemsetAsync(h->w_tcut.p, 0, (size_t)nc, s));
                gh::launch_flag_cut_ties(s, h->w_dist.as<float>(), q_stride, h->w_qtotal.as<int>(), nc, k, h->w_cand_dis.as<float>(),
                                         h->w_cand_pos.as<int>(), nullptr, h->w_tcut.as<uint8_t>(), 0, 1);
                tf.cut = h->w_tcut.as<uint8_t>();
                tf.count = h->w_tlist.as<int>();
                tf.list = h->w_tlist.as<int>() + 1;
                tf.stats = h->d_tie_stats;
            }
            gh::launch_finalize_norank(s, h->w_cand_dis.as<float>(), h->w_cand_ids.as<int64_t>(), nc, k, k, p->min_score,
                                       p->max_score, neutral, d_distances + (size_t)q0 * k, d_labels + (size_t)q0 * k,
                                       ties ? &tf : nullptr);
            if (ties) {
                gh::TieReplayArgs tr;
                flat_tie_args(h, &tr, l2, nc, q_stride, P, k, xq, h->w_cand_dis.as<float>(), h->w_cand_ids.as<int64_t>(),
                              d_distances + (size_t)q0 * k, d_labels + (size_t)q0 * k);
                tr.list = tf.list;
                tr.count = tf.count;
                gh::launch_tie_replay(s, l2, tr);
            }
        }
        GH_CHECK(h, hipEventRecord(h->rd_ev[ver], s));
        h->rd_set[ver] = true;
        h->last_nq = nc;
    }
    h->last_P = P;
    h->last_R = k;
    GH_CHECK(h, hipGetLastError());
    return GAMMA_HIP_OK;
}

// ---- flat ------------------------------------------------------------------------------
int flat_search_device_locked(H* h, const gamma_hip_search_params* p, int nq, const float* d_x, int k,
                              float* d_distances, int64_t* d_labels) {
    if (h->raw_sparse) return fail(h, GAMMA_HIP_EUNSUPPORTED, "the raw store holds this shard's rows only (gamma_hip_raw_put)");
    GH_TRY(check_params(h, p, nq, k));
    gamma_hip_search_params pp;   // (the chunked paths run for k + 1 results: k = 4096, the ABI's largest, is beyond the mode)
    GH_TRY(resolve_ties(h, p, &pp, k + 1 <= gh::tie_replay_max_k() && h->nraw < ((int64_t)1 << 31),
                        "exact_ties = 1 with k = 4096 or 2^31 rows (flat search)"));
    p = &pp;
    if (!h->d_raw && h->nraw > 0) return fail(h, GAMMA_HIP_EINVAL, "raw store not initialised");
    if (h->raw_d <= 0) return fail(h, GAMMA_HIP_EINVAL, "raw store not initialised");
    if (k <= 0 || nq == 0) return GAMMA_HIP_OK;
    GH_CHECK(h, hipSetDevice(h->device));
    const bool l2 = p->metric == GAMMA_HIP_METRIC_L2;
    const float neutral = l2 ? 3.402823466e+38f : -3.402823466e+38f;
    const float sentinel = l2 ? INFINITY : -INFINITY;
    const int d = h->raw_d;
    const int64_t N = h->nraw;
    hipStream_t s = h->stream;
    gh::FilterDesc filt;
    GH_TRY(build_filter(h, p, &filt, nullptr, (int64_t)nq * N));
    {   // small calls: the whole store as ONE row chunk, then the small-batch chains' selection -- three or four launches
        // instead of three per 65536 rows (1 M x 128, one query: 49 launches, 1.1 ms)
        static const bool off = getenv("GAMMA_HIP_NO_SMALL_PATH") != nullptr;
        const int64_t stride = (N + 3) & ~(int64_t)3;
        if (!off && h->small_path && nq <= 64 && k <= 1024 && N >= 1 && N <= ((int64_t)1 << 22) && !h->profile &&
            (size_t)nq * stride * sizeof(float) <= std::min<size_t>(h->dist_budget_bytes, (size_t)1 << 30)) {
            GH_TRY(replay_join(h));
            GH_CHECK(h, h->w_dist.ensure((size_t)nq * stride * sizeof(float)));
            GH_CHECK(h, h->w_cand_pos.ensure((size_t)nq * k * sizeof(int)));
            GH_CHECK(h, h->w_cand_dis.ensure((size_t)nq * k * sizeof(float)));
            GH_CHECK(h, h->w_cand_ids.ensure((size_t)nq * k * sizeof(int64_t)));
            StageScope t(h, GAMMA_HIP_STAGE_FLAT);
            gh::launch_pairwise_filtered(s, l2, d_x, nq, d, h->d_raw, N, h->w_dist.as<float>(), stride, filt, p->min_score,
                                         p->max_score, 0);
            int smax = h->small_presel > 0 ? h->small_presel : (N > 16384 ? (int)std::min<int64_t>(64, (N + 16383) / 16384) : 0);
            if (smax > 0) {
                GH_CHECK(h, h->w_selv.ensure((size_t)nq * smax * k * sizeof(float)));
                GH_CHECK(h, h->w_selp.ensure(((size_t)nq * smax * k + (size_t)nq * smax) * sizeof(int)));   // + a cut flag per slice
            }
            // exact ties: a query with equal distances at the k cut or among its k results is replayed through the
            // reference's heap over the whole row (gamma_index_flat.cc:118-300: heap_pop + heap_push in vid order)
            gh::TieReplayArgs tr;
            const bool ties = tie_on(p) && k <= gh::tie_small_max_k();   // (this path's gate: k <= 1024)
            if (ties) {
                flat_tie_args(h, &tr, l2, nq, stride, 0, k, d_x, h->w_cand_dis.as<float>(), h->w_cand_ids.as<int64_t>(), d_distances,
                              d_labels, (int)N);
                tr.d = d;
            }
            gh::launch_small_tail(s, l2, h->w_dist.as<float>(), stride, nullptr, nq, k, 0, nullptr, nullptr, nullptr, nullptr,
                                  h->w_cand_dis.as<float>(), h->w_cand_pos.as<int>(), h->w_cand_ids.as<int64_t>(), 0, d_x, d,
                                  h->d_raw, N, k, p->min_score, p->max_score, neutral, d_distances, d_labels, smax,
                                  smax ? h->w_selv.as<float>() : nullptr, smax ? h->w_selp.as<int>() : nullptr, (int)N,
                                  ties ? &tr : nullptr, h->d_tie_stats);
            GH_CHECK(h, hipGetLastError());
            return GAMMA_HIP_OK;
        }
    }
    // query chunks x row chunks so the distance slab stays inside the budget
    // The matrix-pipe filter (flat_mfma.hip: bf16 hi / lo products with a proven margin, exact distances of the ~k survivors
    // per query and pass) takes the passes behind the first row chunk when the shape has a variant; the first chunk -- scored
    // exactly for every query, the only part still on the vector ALU -- is then 16384 rows instead of 65536.
    const bool mfma_filter = k <= 256 && N < ((int64_t)1 << 32) && gh::flat_filter_supported(nq, d, N);
    // (long rows: the first chunk's exact kernel runs at ~1.3 TFLOP/s -- 1024 rows of d = 768 instead of 16384: C5 flat, 1 M x 768,
    //  1024 queries: 15.6 / 16.1 / 16.8 / 18.4 ms per call with a first chunk of 2^10 / 2^11 / 2^12 / 2^13 rows)
    static const int first_log2 = getenv("GAMMA_HIP_FLAT_FIRST_LOG2") ? atoi(getenv("GAMMA_HIP_FLAT_FIRST_LOG2")) : 0;
    const int fl2 = (first_log2 >= 8 && first_log2 <= 16 && mfma_filter) ? first_log2 : (mfma_filter ? (d > 128 ? 10 : 14) : 16);
    int64_t rows_chunk = std::max<int64_t>(256, std::min<int64_t>(N, (int64_t)1 << fl2));
    rows_chunk = (rows_chunk + 255) / 256 * 256;
    int qc = (int)std::max<int64_t>(1, std::min<int64_t>(nq, (int64_t)(h->dist_budget_bytes / (rows_chunk * sizeof(float)))));
    const int nchunks = (int)std::max<int64_t>(1, (N + rows_chunk - 1) / rows_chunk);
    // exact ties: the paths below order results on (distance, row id); the reference's heap (heap_pop + heap_push in
    // vid order, heap_reorder: gamma_index_flat.cc:118-300) orders equal distances its own way and decides which members
    // of a tie at the cut stay.  Run for k + 1 results, list the queries with two equal distances among them and replay
    // those through the heap over a freshly computed distance row (k_tie_replay).
    const int k_out = k;
    const bool ties = tie_on(p) && N > 0;
    if (ties) k = k + 1;
    // Overlapping callers (gamma_hip_flat_search_device_wait): this call's tie replay goes to the side stream when the call is one
    // query chunk; a call that finds a FLAT replay pending does not wait for it -- it works in the other bank of the buffers the
    // replay reads and waits, in stream order, only for the replay that last read that bank.
    const bool defer_flat = h->defer_now && h->side2 != nullptr && ties && qc >= nq;
    if (h->replay_pending) {
        if (defer_flat && h->replay_is_flat) {
            DevBuf* cur[5] = {&h->w_dist, &h->w_flog, &h->w_tlist, &h->w_cand_dis, &h->w_cand_ids};
            for (int i = 0; i < 5; i++) std::swap(*cur[i], h->fbank[i]);
            h->flat_bank ^= 1;
            if (h->bank_used[h->flat_bank]) GH_CHECK(h, hipStreamWaitEvent(s, h->ev_bank[h->flat_bank], 0));
            // (replay_pending stays: whoever comes next and cannot switch banks joins)
        } else {
            GH_TRY(replay_join(h));
        }
    }
    auto flat_replay = [&](const gh::TieReplayArgs& tr) -> int {
        if (!defer_flat) {
            gh::launch_tie_replay(s, l2, tr);
            return GAMMA_HIP_OK;
        }
        const int b = h->flat_bank;
        if (!h->ev_bank[b]) GH_CHECK(h, hipEventCreateWithFlags(&h->ev_bank[b], hipEventDisableTiming));
        GH_CHECK(h, hipEventRecord(h->ev_rfork, s));
        GH_CHECK(h, hipStreamWaitEvent(h->side2, h->ev_rfork, 0));
        gh::launch_tie_replay(h->side2, l2, tr);
        GH_CHECK(h, hipEventRecord(h->ev_rdone, h->side2));
        GH_CHECK(h, hipEventRecord(h->ev_bank[b], h->side2));
        h->bank_used[b] = true;
        h->replay_pending = true;
        h->replay_is_flat = true;
        return GAMMA_HIP_OK;
    };
    GH_CHECK(h, h->w_dist.ensure((size_t)qc * rows_chunk * sizeof(float)));
    GH_CHECK(h, h->w_part_v.ensure((size_t)qc * nchunks * k * sizeof(float)));
    GH_CHECK(h, h->w_part_i.ensure((size_t)qc * nchunks * k * sizeof(int64_t)));
    GH_CHECK(h, h->w_selv.ensure((size_t)qc * k * sizeof(float)));
    GH_CHECK(h, h->w_selp.ensure((size_t)qc * k * sizeof(int)));
    GH_CHECK(h, h->w_cand_pos.ensure((size_t)qc * k * sizeof(int)));
    GH_CHECK(h, h->w_cand_dis.ensure((size_t)qc * k * sizeof(float)));
    if (ties) {
        GH_CHECK(h, h->w_fD.ensure((size_t)qc * k * sizeof(float)));
        GH_CHECK(h, h->w_fI.ensure((size_t)qc * k * sizeof(int64_t)));
        GH_CHECK(h, h->w_tlist.ensure(((size_t)qc + 1) * sizeof(int)));
        GH_CHECK(h, h->w_cand_ids.ensure((size_t)qc * k * sizeof(int64_t)));
    }
    // where the paths write their [nc][k] result of query chunk q0
    auto outD = [&](int q0) { return ties ? h->w_fD.as<float>() : d_distances + (size_t)q0 * k; };
    auto outI = [&](int q0) { return ties ? h->w_fI.as<int64_t>() : d_labels + (size_t)q0 * k; };
    StageScope t(h, GAMMA_HIP_STAGE_FLAT);
    // the reference's loop: every row, one query at a time, a k-heap (gamma_index_flat.cc:118-300).
    // Here: distance slab of one row chunk -> per-chunk top-k -> merge of the chunks' tables.
    auto unbounded = [&](int q0, int nc) -> int {
        const float* xq = d_x + (size_t)q0 * d;
        for (int c = 0; c < nchunks; c++) {
            const int64_t r0 = (int64_t)c * rows_chunk;
            const int64_t nr = std::min<int64_t>(rows_chunk, N - r0);
            gh::launch_pairwise_filtered(s, l2, xq, nc, d, h->d_raw + r0 * d, nr, h->w_dist.as<float>(),
                                         rows_chunk, filt, p->min_score, p->max_score, r0);
            // per-chunk top-k: values + positions relative to the chunk
            gh::launch_select_topk(s, l2, h->w_dist.as<float>(), rows_chunk, nullptr, (int)nr, (int)nr, nc, k,
                                   h->w_cand_dis.as<float>(), h->w_cand_pos.as<int>());
            // scatter into the partial table [q][chunk][k] with global ids
            // (reuse finalize_topk: labels = pos, then offset by r0 on the fly below)
            gh::launch_finalize_topk(s, h->w_cand_dis.as<float>(), h->w_cand_pos.as<int>(), nc, k, nullptr,
                                     0, r0, sentinel, h->w_selv.as<float>(),
                                     h->w_part_i.as<int64_t>() + (size_t)c * nc * k);
            GH_CHECK(h, hipMemcpyAsync(h->w_part_v.as<float>() + (size_t)c * nc * k, h->w_selv.p,
                                       (size_t)nc * k * sizeof(float), hipMemcpyDeviceToDevice, s));
        }
        // merge: layout [chunk][q][k] == the sharded layout [shard][nq][R]
        GH_CHECK(h, h->w_m_dis.ensure((size_t)nc * nchunks * k * sizeof(float)));
        GH_CHECK(h, h->w_m_ids.ensure((size_t)nc * nchunks * k * sizeof(int64_t)));
        gh::launch_gather_shards(s, h->w_part_v.as<float>(), h->w_part_i.as<int64_t>(), nchunks, nc, k,
                                 h->w_m_dis.as<float>(), h->w_m_ids.as<int64_t>(), sentinel);
        gh::launch_select_topk(s, l2, h->w_m_dis.as<float>(), (int64_t)nchunks * k, nullptr, nchunks * k,
                               nchunks * k, nc, k, h->w_selv.as<float>(), h->w_selp.as<int>());
        gh::launch_finalize_topk(s, h->w_selv.as<float>(), h->w_selp.as<int>(), nc, k,
                                 h->w_m_ids.as<int64_t>(), (int64_t)nchunks * k, 0, neutral,
                                 outD(q0), outI(q0));
        return GAMMA_HIP_OK;
    };
    // Running bound: only the first chunk goes through a distance slab.  Its k-th best bounds the
    // answer; the remaining rows are scored in passes that double the rows seen so far, each pass
    // appending only the distances within the current bound to the query's candidate list (about k
    // per pass and query) and ending with a compaction that tightens the bound.  The k smallest
    // (distance, row id) items are the same either way.  A list that overflows (rows arriving in
    // improving order) is detected and the call redone without a bound.
    const int cap = gh::flat_list_cap();
    int log_nsl = 1;   // slices of the log: one per pass of the running bound (+ slice 0, the first chunk's slab)
    // pass sizes: every pass scores (growth - 1) x the rows scored so far -- with a bound from r rows a pass over g r more
    // lets ~k ln(1 + g) new candidates per query through; fewer, larger passes pay fewer fixed launches (bounds, filter
    // ramp, exact, compact: ~100 us each at C2)
    // (C2, ms per call at growth 2 / 3 / 4 / 6: 2.26 / 2.19 / 2.15 / 2.24)
    static const int flat_growth = getenv("GAMMA_HIP_FLAT_GROWTH") ? std::max(2, atoi(getenv("GAMMA_HIP_FLAT_GROWTH"))) : 4;
    auto pass_rows = [&](int64_t r) { return std::min<int64_t>((int64_t)(flat_growth - 1) * r, N - r); };
    for (int64_t r = rows_chunk; r < N; r += pass_rows(r)) log_nsl++;
    auto bounded = [&](int q0, int nc, bool* redo) -> int {
        const float* xq = d_x + (size_t)q0 * d;
        GH_CHECK(h, h->w_flat_cand.ensure((size_t)nc * cap * sizeof(unsigned long long)));
        // tau[nc] | cnt[nc * CS] | overflow -- the counters one 128-byte line apart: ~140 appends per query and pass are
        // returning atomics, and 32 queries' counters on one line serialised 4500 of them (C2 exact passes 49 / 66 / 84 -> 30 / 31 / 37 us)
        constexpr int CS = 32;
        GH_CHECK(h, h->w_flat_meta.ensure((size_t)(nc + (size_t)nc * CS + 1) * sizeof(int)));
        uint32_t* tau = h->w_flat_meta.as<uint32_t>();
        int* cnt = h->w_flat_meta.as<int>() + nc;
        int* over = cnt + (size_t)nc * CS;
        gh::FlatEmit em{tau, h->w_flat_cand.as<unsigned long long>(), cnt, cap, CS};
        GH_CHECK(h, hipMemsetAsync(over, 0, sizeof(int), s));
        gh::FlatLog lg;
        if (ties) {   // what every pass appends is kept for the replay (slice `pass` of the query)
            GH_CHECK(h, h->w_flog.ensure((size_t)nc * log_nsl * cap * sizeof(unsigned long long) +
                                         (size_t)nc * (log_nsl + 1) * sizeof(int)));
            lg.items = h->w_flog.as<unsigned long long>();
            lg.cnt = reinterpret_cast<int*>(lg.items + (size_t)nc * log_nsl * cap);
            lg.kept = lg.cnt + (size_t)nc * log_nsl;
            lg.nsl = log_nsl;
            GH_CHECK(h, hipMemsetAsync(lg.cnt, 0, (size_t)nc * log_nsl * sizeof(int), s));
        }
        gh::launch_pairwise_filtered(s, l2, xq, nc, d, h->d_raw, rows_chunk, h->w_dist.as<float>(), rows_chunk, filt,
                                     p->min_score, p->max_score, 0);
        gh::launch_select_topk(s, l2, h->w_dist.as<float>(), rows_chunk, nullptr, (int)rows_chunk, (int)rows_chunk,
                               nc, k, h->w_cand_dis.as<float>(), h->w_cand_pos.as<int>());
        gh::launch_flat_init(s, l2, h->w_cand_dis.as<float>(), h->w_cand_pos.as<int>(), nc, k, 0, em, tau, lg.kept);
        const bool mf = mfma_filter && gh::flat_filter_supported(nc, d, N);
        const int64_t pcap = gh::flat_filter_pair_cap(nc);
        float* bnd = nullptr;
        int* npairs = nullptr;
        if (mf) {   // once per query chunk: norms and the queries' bf16 hi / lo image
            GH_CHECK(h, h->w_xn.ensure((size_t)nc * sizeof(float)));
            GH_CHECK(h, h->w_fq.ensure(gh::flat_filter_query_image_bytes(nc, d)));
            GH_CHECK(h, h->w_fraw.ensure((size_t)pcap * 8));
            GH_CHECK(h, h->w_frcnt.ensure(gh::flat_filter_counter_bytes() + gh::flat_filter_bounds_bytes(nc)));   // pair counters | bounds
            npairs = h->w_frcnt.as<int>();
            bnd = reinterpret_cast<float*>(h->w_frcnt.as<char>() + gh::flat_filter_counter_bytes());
            gh::launch_row_norms(s, xq, nc, d, h->w_xn.as<float>());
            gh::launch_flat_prep_queries(s, xq, nc, d, h->w_fq.p);
        }
        for (int64_t r = rows_chunk; r < N;) {
            const int64_t nr = pass_rows(r);
            if (mf) {
                GH_CHECK(h, hipMemsetAsync(npairs, 0, (size_t)gh::flat_filter_counter_bytes(), s));
                gh::launch_flat_filter(s, l2, d, h->w_fq.p, h->w_xn.as<float>(), tau, bnd, nc, h->d_raw + r * d, nr, r,
                                       h->w_fraw.p, npairs, pcap);
                gh::launch_flat_exact(s, l2, h->w_fraw.p, npairs, pcap, xq, nc, d, h->d_raw, filt, p->min_score, p->max_score,
                                      em, over);
            } else {
                gh::launch_pairwise_emit(s, l2, xq, nc, d, h->d_raw + r * d, nr, filt, p->min_score, p->max_score, r, em);
            }
            lg.pass++;
            gh::launch_flat_compact(s, nc, k, em, tau, over, ties ? &lg : nullptr);
            r += nr;
        }
        gh::launch_flat_final(s, l2, nc, k, em, neutral, outD(q0), outI(q0));
        GH_CHECK(h, hipGetLastError());
        // (whether a list overflowed -- then the call is redone without a bound -- is read back by the caller AFTER it has
        //  enqueued the tie phase: the device never waits for the host)
        *redo = false;
        return GAMMA_HIP_OK;
    };
    auto overflowed = [&](int nc, bool* yes) -> int {
        int h_over = 0;
        GH_CHECK(h, hipMemcpyAsync(&h_over, h->w_flat_meta.as<int>() + nc + (size_t)nc * 32, sizeof(int), hipMemcpyDeviceToHost, s));
        GH_CHECK(h, hipStreamSynchronize(s));
        *yes = h_over != 0;
        return GAMMA_HIP_OK;
    };
    for (int q0 = 0; q0 < nq; q0 += qc) {
        const int nc = std::min(qc, nq - q0);
        if (N == 0) {
            // nothing to scan: all-empty result
            GH_CHECK(h, hipMemsetAsync(h->w_selp.p, 0xff, (size_t)nc * k * sizeof(int), s));
            gh::launch_finalize_topk(s, h->w_selv.as<float>(), h->w_selp.as<int>(), nc, k, nullptr, 0, 0,
                                     neutral, outD(q0), outI(q0));
            continue;
        }
        auto tie_phase = [&](bool redo) -> int {
            int* count = h->w_tlist.as<int>();
            int* list = count + 1;
            GH_CHECK(h, hipMemsetAsync(count, 0, sizeof(int), s));
            gh::launch_flat_take_flag(s, h->w_fD.as<float>(), h->w_fI.as<int64_t>(), nc, k_out, d_distances + (size_t)q0 * k_out,
                                      d_labels + (size_t)q0 * k_out, list, count, h->d_tie_stats);
            gh::TieReplayArgs tr;
            flat_tie_args(h, &tr, l2, nc, rows_chunk, 0, k_out, d_x + (size_t)q0 * d, h->w_cand_dis.as<float>(),
                          h->w_cand_ids.as<int64_t>(), d_distances + (size_t)q0 * k_out, d_labels + (size_t)q0 * k_out, (int)N);
            tr.d = d;
            tr.list = list;
            tr.count = count;
            if (!redo) {
                // the running bound: the stream the heap saw = the first row chunk (its slab is intact) + what every
                // later pass appended to the query's list, in row order
                tr.G = (int)rows_chunk;
                tr.always_sliced = 1;
                tr.surv = h->w_flog.as<unsigned long long>();
                tr.gcnt = reinterpret_cast<const int*>(tr.surv + (size_t)nc * log_nsl * cap);
                tr.nsl = log_nsl;
                tr.slice_cap = cap;
                GH_TRY(flat_replay(tr));
            } else if (nchunks == 1) {
                GH_TRY(flat_replay(tr));   // the one slab holds every row
            } else {
                GH_TRY(replay_join(h));
                // several row chunks through one slab: the rows of the flagged queries are computed again
                int nflag = 0;
                GH_CHECK(h, hipMemcpyAsync(&nflag, count, sizeof(int), hipMemcpyDeviceToHost, s));
                GH_CHECK(h, hipStreamSynchronize(s));
                const int64_t stride = (N + 3) & ~(int64_t)3;
                const int fcap = (int)std::max<int64_t>(1, std::min<int64_t>(1024, (int64_t)(h->dist_budget_bytes / ((size_t)stride * sizeof(float)))));
                for (int f0 = 0; f0 < nflag; f0 += fcap) {
                    const int nf = std::min(fcap, nflag - f0);
                    GH_CHECK(h, h->w_fx.ensure((size_t)nf * d * sizeof(float)));
                    GH_CHECK(h, h->w_fslab.ensure((size_t)nf * stride * sizeof(float)));
                    gh::launch_gather_rows(s, d_x + (size_t)q0 * d, list + f0, nf, d, h->w_fx.as<float>());
                    gh::launch_pairwise_filtered(s, l2, h->w_fx.as<float>(), nf, d, h->d_raw, N, h->w_fslab.as<float>(), stride, filt,
                                                 p->min_score, p->max_score, 0);
                    gh::TieReplayArgs t2 = tr;
                    t2.nq = nf;
                    t2.slab = h->w_fslab.as<float>();
                    t2.q_stride = stride;
                    t2.list = list + f0;
                    t2.compact_rows = 1;
                    gh::launch_tie_replay(s, l2, t2);
                }
            }
            return GAMMA_HIP_OK;
        };
        bool redo = true;
        if (!h->flat_no_bound && k <= 256 && N > rows_chunk && N < ((int64_t)1 << 32) &&
            (gh::pairwise_can_emit(nc, d, N - rows_chunk) || (mfma_filter && gh::flat_filter_supported(nc, d, N)))) {
            GH_TRY(bounded(q0, nc, &redo));
            if (ties) GH_TRY(tie_phase(false));   // enqueued before the host learns whether a list overflowed
            if (defer_flat && h->flat_over_dst) {
                // (the caller reads the word after the call's completion event, with the handle free: no host wait under the lock)
                GH_CHECK(h, hipMemcpyAsync(h->flat_over_dst, h->w_flat_meta.as<int>() + nc + (size_t)nc * 32, sizeof(int),
                                           hipMemcpyDeviceToHost, s));
                continue;
            }
            GH_TRY(overflowed(nc, &redo));
            if (!redo) continue;
            GH_TRY(replay_join(h));   // (a replay on the side stream writes the rows that are about to be redone)
        }
        GH_TRY(unbounded(q0, nc));
        if (ties) GH_TRY(tie_phase(true));
    }
    GH_CHECK(h, hipGetLastError());
    return GAMMA_HIP_OK;
}

int flat_search_host_locked(gamma_hip_index* h, const gamma_hip_search_params* p, int nq, const float* x, int k,
                                  float* distances, int64_t* labels) {
    SearchLock lk(h);
    GH_TRY(check_params(h, p, nq, k));
    if (h->raw_d <= 0) return fail(h, GAMMA_HIP_EINVAL, "raw store not initialised");
    if (nq > 0 && k > 0 && (!x || !distances || !labels)) return fail(h, GAMMA_HIP_EINVAL, "null buffer");
    return host_search(h, nq, h->raw_d, x, k, distances, labels, [&](const float* dx, float* dd, int64_t* dl) {
        return flat_search_device_locked(h, p, nq, dx, k, dd, dl);
    }, true, &lk);
}

int ivfpq_search_host_locked(gamma_hip_index* h, const gamma_hip_search_params* p, int nq, const float* x,
                                   int k, float* distances, int64_t* labels) {
    SearchLock lk(h);
    GH_TRY(ivfpq_check(h, p, nq, k));
    if (nq > 0 && k > 0 && (!x || !distances || !labels)) return fail(h, GAMMA_HIP_EINVAL, "null buffer");
    return host_search(h, nq, h->d, x, k, distances, labels, [&](const float* dx, float* dd, int64_t* dl) {
        return ivfpq_search_device_locked(h, p, nq, dx, k, dd, dl);
    }, true, &lk);
}

// A LARGE host-buffer call (the plugin's Search with thousands of queries) when several client threads share the
// handle: the reference's Search is re-entrant (tests/test.h:1033-1062).  The call's queries are uploaded on a stream of
// their own into one of two staging slots BEFORE the search lock is taken (the previous caller's kernels are running);
// under the lock the search is enqueued with its tie replay on the side stream, the results follow the replay down into
// pinned memory, and the lock is released: the caller waits for ITS completion event while the next caller's coarse
// quantizer / tables / scan run beside this call's replay and copies.  One caller alone loses nothing.
int ivfpq_search_host_overlap(gamma_hip_index* h, const gamma_hip_search_params* p, int nq, const float* x, int k,
                              float* distances, int64_t* labels) {
    const int d = h->d;
    H::HostSlot* sl = nullptr;
    {
        std::unique_lock<std::mutex> g(h->hs_mu);
        h->hs_cv.wait(g, [&] { return !h->hslot[0].busy || !h->hslot[1].busy; });
        sl = !h->hslot[0].busy ? &h->hslot[0] : &h->hslot[1];
        sl->busy = true;
    }
    struct Release {
        H* h; H::HostSlot* sl;
        ~Release() {
            { std::lock_guard<std::mutex> g(h->hs_mu); sl->busy = false; }
            h->hs_cv.notify_one();
        }
    } release{h, sl};
    GH_CHECK(h, hipSetDevice(h->device));
    const size_t bx = (size_t)nq * d * sizeof(float), bd = (size_t)nq * k * sizeof(float), bi = (size_t)nq * k * sizeof(int64_t);
    // (the queries through pinned memory too -- a CPU copy, then a true asynchronous upload -- measured slower: 10.66 against
    //  10.90 M q/s with two callers, 7.5 against 8.2 with one)
    const size_t off_i = 0, off_d = (bi + 63) & ~(size_t)63, need = off_d + bd;
    GH_CHECK(h, sl->x.ensure(bx));
    GH_CHECK(h, sl->D.ensure(bd));
    GH_CHECK(h, sl->I.ensure(bi));
    if (need > sl->pin_bytes) {
        if (sl->pin) (void)hipHostFree(sl->pin);
        sl->pin = nullptr;
        sl->pin_bytes = 0;
        GH_CHECK(h, hipHostMalloc(&sl->pin, need + need / 4, hipHostMallocDefault));
        sl->pin_bytes = need + need / 4;
    }
    if (!sl->ev_up) GH_CHECK(h, hipEventCreateWithFlags(&sl->ev_up, hipEventDisableTiming));
    if (!sl->ev_done) GH_CHECK(h, hipEventCreateWithFlags(&sl->ev_done, hipEventDisableTiming));
    {
        std::lock_guard<std::mutex> g(h->hs_mu);
        if (!h->up_stream) GH_CHECK(h, hipStreamCreateWithFlags(&h->up_stream, hipStreamNonBlocking));
    }
    // (pageable source: the runtime stages it; the thread is held here while the copy engine works -- beside whatever
    //  the search stream is running for the previous caller)
    char* pin = static_cast<char*>(sl->pin);
    GH_CHECK(h, hipMemcpyAsync(sl->x.p, x, bx, hipMemcpyHostToDevice, h->up_stream));
    GH_CHECK(h, hipEventRecord(sl->ev_up, h->up_stream));
    {
        SearchLock lk(h);
        GH_TRY(ivfpq_check(h, p, nq, k));
        GH_CHECK(h, hipStreamWaitEvent(h->stream, sl->ev_up, 0));
        h->defer_now = h->side2 != nullptr;
        const int rc = ivfpq_search_device_locked(h, p, nq, sl->x.as<float>(), k, sl->D.as<float>(), sl->I.as<int64_t>());
        h->defer_now = false;
        if (rc != GAMMA_HIP_OK) return rc;
        hipStream_t ts = h->replay_pending ? h->side2 : h->stream;   // behind this call's replay, if it has one
        GH_CHECK(h, hipMemcpyAsync(pin + off_d, sl->D.p, bd, hipMemcpyDeviceToHost, ts));
        GH_CHECK(h, hipMemcpyAsync(pin + off_i, sl->I.p, bi, hipMemcpyDeviceToHost, ts));
        GH_CHECK(h, hipEventRecord(sl->ev_done, ts));
    }
    if (hipEventSynchronize(sl->ev_done) != hipSuccess) return fail(h, GAMMA_HIP_EDEVICE, "waiting for the call's completion event");
    std::memcpy(distances, pin + off_d, bd);
    std::memcpy(labels, pin + off_i, bi);
    return GAMMA_HIP_OK;
}

}  // namespace ghi

using namespace ghi;

extern "C" {

/* ---- search ----------------------------------------------------------------------------- */
int gamma_hip_ivfpq_search_device(gamma_hip_index* h, const gamma_hip_search_params* p, int nq,
                                  const float* d_x, int k, float* d_distances, int64_t* d_labels) {
    if (!h) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    h->defer_now = h->defer_replay;
    const int rc = ivfpq_search_device_locked(h, p, nq, d_x, k, d_distances, d_labels);
    h->defer_now = false;
    return rc;
}

int gamma_hip_ivfpq_search_device_wait(gamma_hip_index* h, const gamma_hip_search_params* p, int nq,
                                       const float* d_x, int k, float* d_distances, int64_t* d_labels) {
    if (!h) return GAMMA_HIP_EINVAL;
    hipEvent_t ev = nullptr;
    int dev = 0;
    {
        SearchLock lk(h);
        dev = h->device;
        GH_CHECK(h, hipSetDevice(h->device));
        const unsigned slot = h->call_seq++ & 3u;
        if (!h->ev_call[slot]) GH_CHECK(h, hipEventCreateWithFlags(&h->ev_call[slot], hipEventDisableTiming));
        ev = h->ev_call[slot];
        h->defer_now = h->side2 != nullptr;   // the replay of this call's flagged queries: side stream, beside the next caller's head
        const int rc = ivfpq_search_device_locked(h, p, nq, d_x, k, d_distances, d_labels);
        h->defer_now = false;
        if (rc != GAMMA_HIP_OK) return rc;
        // everything of the call precedes the replay's fork; the replay (if one is pending: this call's) ends the call
        GH_CHECK(h, hipEventRecord(ev, h->replay_pending ? h->side2 : h->stream));
    }
    // the handle is free from here on: the next caller enqueues while this one waits for its own call
    (void)dev;
    const hipError_t e = hipEventSynchronize(ev);
    return e == hipSuccess ? GAMMA_HIP_OK : GAMMA_HIP_EDEVICE;
}

int gamma_hip_ivfflat_search_device(gamma_hip_index* h, const gamma_hip_search_params* p, int nq, const float* d_x,
                                    int k, float* d_distances, int64_t* d_labels) {
    if (!h) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    GH_TRY(replay_join(h));
    return ivfflat_search_device_locked(h, p, nq, d_x, k, d_distances, d_labels);
}

int gamma_hip_ivfflat_search(gamma_hip_index* h, const gamma_hip_search_params* p, int nq, const float* x, int k,
                             float* distances, int64_t* labels) {
    if (!h) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    GH_TRY(check_params(h, p, nq, k));
    if (!h->ivf_init || !h->ivfflat) return fail(h, GAMMA_HIP_EINVAL, "ivfflat not initialised");
    if (nq > 0 && k > 0 && (!x || !distances || !labels)) return fail(h, GAMMA_HIP_EINVAL, "null buffer");
    return host_search(h, nq, h->d, x, k, distances, labels, [&](const float* dx, float* dd, int64_t* dl) {
        return ivfflat_search_device_locked(h, p, nq, dx, k, dd, dl);
    }, true, &lk);
}

int gamma_hip_ivfpq_search(gamma_hip_index* h, const gamma_hip_search_params* p, int nq, const float* x,
                           int k, float* distances, int64_t* labels) {
    if (!h) return GAMMA_HIP_EINVAL;
    if (h->combine && p && nq > 0 && nq <= COMB_MAX_NQ && k > 0 && x && distances && labels && h->ivf_init && h->d > 0 &&
        (!p->has_range || (p->n_range >= 0 && p->n_range <= gh::kMaxRange && (p->n_range == 0 || p->range))) &&
        p->n_field >= 0 && p->n_field <= gh::kMaxField && (p->n_field == 0 || p->field) &&
        p->n_term >= 0 && p->n_term <= gh::kMaxTerm && (p->n_term == 0 || p->term))
        return combined_search(h, p, nq, x, k, distances, labels);
    // large calls: staged so that concurrent callers overlap (ivfpq_search_host_overlap); GAMMA_HIP_NO_HOST_OVERLAP=1: the plain path
    static const bool no_overlap = getenv("GAMMA_HIP_NO_HOST_OVERLAP") != nullptr;
    // (a caller that finds the handle idle takes the plain path: results straight into its buffers, 1.89 against 1.96 ms per
    //  16384-query call; one that finds another large call in flight takes the staged path -- two alternating callers are
    //  both on it from their second call on)
    struct InFlight {
        std::atomic<int>& n;
        int before;
        explicit InFlight(std::atomic<int>& c) : n(c), before(c.fetch_add(1)) {}
        ~InFlight() { n.fetch_sub(1); }
    };
    const bool big = p && nq > 0 && k > 0 && x && distances && labels && h->ivf_init && !h->ivfflat && h->d > 0 &&
                     (size_t)nq * h->d * sizeof(float) > ((size_t)1 << 20);
    if (!big) return ivfpq_search_host_locked(h, p, nq, x, k, distances, labels);
    InFlight fl(h->big_calls_in_flight);
    // (sticky for 50 ms: alternating callers are now and then both between two calls)
    const int64_t now_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
    if (fl.before > 0) h->big_calls_overlap_seen_ns.store(now_ns, std::memory_order_relaxed);
    const bool recent = now_ns - h->big_calls_overlap_seen_ns.load(std::memory_order_relaxed) < 50000000LL;
    if (!no_overlap && h->side2 && (fl.before > 0 || recent)) return ivfpq_search_host_overlap(h, p, nq, x, k, distances, labels);
    return ivfpq_search_host_locked(h, p, nq, x, k, distances, labels);
}

int gamma_hip_ivfpq_last_stages(gamma_hip_index* h, float* coarse_dis, int64_t* coarse_idx,
                                float* recall_dis, int64_t* recall_ids) {
    if (!h) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    GH_TRY(replay_join(h));
    const int nq = h->last_nq, P = h->last_P, R = h->last_R;
    if (nq <= 0) return fail(h, GAMMA_HIP_EINVAL, "no previous search");
    GH_CHECK(h, hipSetDevice(h->device));
    GH_CHECK(h, hipStreamSynchronize(h->stream));
    if (coarse_dis) GH_CHECK(h, hipMemcpy(coarse_dis, h->w_coarse_dis.p, (size_t)nq * P * sizeof(float), hipMemcpyDeviceToHost));
    if (coarse_idx) {
        std::vector<int> tmp((size_t)nq * P);
        GH_CHECK(h, hipMemcpy(tmp.data(), h->w_probe.p, tmp.size() * sizeof(int), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < tmp.size(); i++) coarse_idx[i] = tmp[i];
    }
    if (recall_dis) GH_CHECK(h, hipMemcpy(recall_dis, h->w_cand_dis.p, (size_t)nq * R * sizeof(float), hipMemcpyDeviceToHost));
    if (recall_ids) GH_CHECK(h, hipMemcpy(recall_ids, h->w_cand_ids.p, (size_t)nq * R * sizeof(int64_t), hipMemcpyDeviceToHost));
    return GAMMA_HIP_OK;
}

int gamma_hip_ivfpq_search_shard(gamma_hip_index* h, const gamma_hip_search_params* p, int nq,
                                 const float* d_x, int k, float* d_recall_dis, int64_t* d_recall_ids) {
    if (!h) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    GH_TRY(replay_join(h));
    GH_TRY(ivfpq_check(h, p, nq, k));
    if (k <= 0 || nq == 0) return GAMMA_HIP_OK;
    if (!d_x || !d_recall_dis || !d_recall_ids) return fail(h, GAMMA_HIP_EINVAL, "null buffer");
    GH_CHECK(h, hipSetDevice(h->device));
    const int R = std::max(p->recall_num, k);
    gh::FilterDesc filt;
    GH_TRY(build_filter(h, p, &filt));
    FiltCtx fc;
    GH_TRY(filt_ctx_single(h, filt, &fc));
    gamma_hip_search_params pp;
    GH_TRY(resolve_ties(h, p, &pp, p->nprobe <= gh::tie_replay_max_probes(), "exact_ties = 1 with nprobe > 1024"));
    if (pp.coarse_mode < 0) pp.coarse_mode = nq < 20 ? 0 : 1;   // decided on the whole call, not per chunk
    if (pp.coarse_mode == 1 && blas_form_not_restated(nq, h->nlist, h->d)) h->blas_unrestated++;
    p = &pp;
    const int chunk = query_chunk(h, nq, p->nprobe);
    for (int q0 = 0; q0 < nq; q0 += chunk) {
        const int nc = std::min(chunk, nq - q0);
        GH_TRY(ivfpq_stage_a(h, p, fc.at(q0), nc, d_x + (size_t)q0 * h->d, R, nullptr, nullptr, /*shard=*/true,
                             d_recall_dis + (size_t)q0 * R, d_recall_ids + (size_t)q0 * R));
        h->last_nq = nc;
    }
    h->last_P = p->nprobe;
    h->last_R = R;
    return GAMMA_HIP_OK;
}

int gamma_hip_ivfpq_coarse_device(gamma_hip_index* h, const gamma_hip_search_params* p, int nq,
                                  const float* d_x, float* d_coarse_dis, int32_t* d_probe) {
    if (!h) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    GH_TRY(replay_join(h));
    GH_TRY(ivfpq_check(h, p, nq, 1));
    if (nq == 0) return GAMMA_HIP_OK;
    if (!d_x || !d_coarse_dis || !d_probe) return fail(h, GAMMA_HIP_EINVAL, "null buffer");
    GH_CHECK(h, hipSetDevice(h->device));
    const int P = p->nprobe;
    gamma_hip_search_params pp;
    GH_TRY(resolve_ties(h, p, &pp, p->nprobe <= gh::tie_replay_max_probes(), "exact_ties = 1 with nprobe > 1024"));
    // the caller resolves coarse_mode -1 on the size of the whole batch; a slice that arrives unresolved decides by itself
    if (pp.coarse_mode < 0) pp.coarse_mode = nq < 20 ? 0 : 1;
    if (pp.coarse_mode == 1 && blas_form_not_restated(nq, h->nlist, h->d)) h->blas_unrestated++;
    p = &pp;
    const int chunk = coarse_chunk(h, nq);
    for (int q0 = 0; q0 < nq; q0 += chunk)
        GH_TRY(ivfpq_coarse(h, p, std::min(chunk, nq - q0), d_x + (size_t)q0 * h->d, d_coarse_dis + (size_t)q0 * P,
                            d_probe + (size_t)q0 * P));
    GH_CHECK(h, hipGetLastError());
    return GAMMA_HIP_OK;
}

static int shard_preassigned(gamma_hip_index* h, const gamma_hip_search_params* p, int nq,
                             const float* d_x, const float* d_coarse_dis,
                             const int32_t* d_probe, int k, float* d_recall_dis,
                             int64_t* d_recall_ids, ghi::BoundXchg* bx) {
    if (!h) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    // (a call with a reduction makes it exactly once, whatever happens: the other shards are waiting in theirs)
    struct ReduceOnce {
        H* h; ghi::BoundXchg* bx; int nq; bool l2;
        ~ReduceOnce() {
            if (bx && !bx->called && bx->d_bound && nq > 0) {
                gh::launch_fill_f32(h->stream, bx->d_bound, nq, l2 ? INFINITY : -INFINITY);
                (void)bx->reduce(h, nq, l2);
            }
        }
    } reduce_once{h, bx, nq, p && p->metric == GAMMA_HIP_METRIC_L2};
    GH_TRY(replay_join(h));
    GH_TRY(ivfpq_check(h, p, nq, k));
    if (k <= 0 || nq == 0) return GAMMA_HIP_OK;
    if (!d_coarse_dis || !d_probe) return fail(h, GAMMA_HIP_EINVAL, "null coarse assignment");
    if (!d_x || !d_recall_dis || !d_recall_ids) return fail(h, GAMMA_HIP_EINVAL, "null buffer");
    if (bx && !bx->d_bound) return fail(h, GAMMA_HIP_EINVAL, "null bound buffer");
    GH_CHECK(h, hipSetDevice(h->device));
    const int R = std::max(p->recall_num, k), P = p->nprobe;
    gamma_hip_search_params pp;
    GH_TRY(resolve_ties(h, p, &pp, P <= gh::tie_replay_max_probes(), "exact_ties = 1 with nprobe > 1024"));
    p = &pp;
    gh::FilterDesc filt;
    GH_TRY(build_filter(h, p, &filt));
    FiltCtx fc;
    GH_TRY(filt_ctx_single(h, filt, &fc));
    int chunk = query_chunk(h, nq, P);
    struct StrideScope {   // the measured stride holds for this call only
        H* h;
        ~StrideScope() { h->q_stride_cap = 0; }
    } stride_scope{h};
    static const bool no_two = getenv("GAMMA_HIP_NO_TWO_PHASE") != nullptr;
    if (chunk < nq || (bx && !no_two)) {
        // A shard scans ~P / W of a query's probes, but the slab stride of the general path is P x the longest list:
        // the budget then cuts the batch (W times the queries of one rank) into many chunks, each with its own launches
        // and its own handful of fallback queries (full-size C4, W = 8: 20 chunks, 5.7 ms of a 37 ms step).  The
        // assignment is on the device: one small kernel measures the longest candidate row of THIS batch over THIS
        // shard's lists, the host reads the one word back (the call is tens of milliseconds long) and sizes the chunks
        // by it.
        GH_CHECK(h, h->w_shard_cut.ensure(std::max<size_t>((size_t)nq, 16)));   // (its first word; the flags come later)
        gh::launch_max_local_total(h->stream, d_probe, nq, P, h->d_list_len, h->d_list_mask, h->nlist, h->w_shard_cut.as<int>());
        int mx[2] = {0, 0};
        GH_CHECK(h, hipMemcpyAsync(mx, h->w_shard_cut.p, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        GH_CHECK(h, hipStreamSynchronize(h->stream));
        h->q_stride_cap = (std::max<int64_t>(mx[0], 1) + 3) & ~(int64_t)3;
        const int64_t by_dist = (int64_t)(h->dist_budget_bytes / ((size_t)h->q_stride_cap * sizeof(float)));
        chunk = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(by_dist, coarse_chunk(h, nq)), nq));
        if (bx && !no_two && chunk < nq && (int64_t)nq <= coarse_chunk(h, nq)) {
            // Two phases need the batch in ONE chunk (one reduction per call).  The slab of a W-rank job's batch -- W x the
            // queries of a rank, each with its longest-case row -- can exceed the general workspace budget (full-size C4, 8
            // shards: ~40 GB against 32); a list shard holds 1 / W of the index, so the memory is there: up to half of what
            // is free now (the workspace stays allocated for the calls that follow).
            size_t free_b = 0, total_b = 0;
            const size_t need = (size_t)nq * (size_t)h->q_stride_cap * sizeof(float);
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need <= h->w_dist.cap + free_b / 2) chunk = nq;
            else (void)hipGetLastError();
        }
        if (bx && !no_two && chunk >= nq) {
            // two phases (ivfpq_stage_a, BoundXchg): the batch runs as ONE chunk, so the reduction is one collective; the scan
            // behind sees a search of P' probes -- the most owned probes any query of the batch has -- all of them dense
            const char* g1e = getenv("GAMMA_HIP_SHARD_G1");   // (read per call: tools sweep it inside one process)
            const int g1_env = g1e ? atoi(g1e) : 2;
            bx->two_phase = true;
            bx->P_in = P;
            bx->G1 = std::max(1, g1_env);
            const int g = bx->G1;
            pp.nprobe = std::max(1, std::min(P, ((std::max(mx[1], 1) + g - 1) / g) * g));
        }
    }
    const int P_in = P;
    const int Pe = pp.nprobe;   // rows of the compacted assignment the scan sees (P unless two phases)
    bool all_cut = true;
    for (int q0 = 0; q0 < nq; q0 += chunk) {
        const int nc = std::min(chunk, nq - q0);
        GH_TRY(ivfpq_stage_a(h, p, fc.at(q0), nc, d_x + (size_t)q0 * h->d, R, d_coarse_dis + (size_t)q0 * P_in,
                             d_probe + (size_t)q0 * P_in, /*shard=*/true, d_recall_dis + (size_t)q0 * R,
                             d_recall_ids + (size_t)q0 * R, (bx && bx->two_phase) ? bx : nullptr));
        h->last_nq = nc;
        if (chunk < nq && h->shard_cut_nq == nc) {   // a call of several chunks: the cut-tie flags of all of them
            GH_CHECK(h, h->w_shard_cut.ensure((size_t)nq));
            GH_CHECK(h, hipMemcpyAsync(h->w_shard_cut.as<uint8_t>() + q0, h->w_tcut.p, (size_t)nc, hipMemcpyDeviceToDevice, h->stream));
        } else if (chunk < nq) {
            all_cut = false;
        }
    }
    h->shard_cut_chunked = chunk < nq && all_cut;
    if (h->shard_cut_chunked) h->shard_cut_nq = nq;
    h->last_P = Pe;
    h->last_R = R;
    return GAMMA_HIP_OK;
}

int gamma_hip_ivfpq_search_shard_preassigned(gamma_hip_index* h, const gamma_hip_search_params* p, int nq,
                                             const float* d_x, const float* d_coarse_dis,
                                             const int32_t* d_probe, int k, float* d_recall_dis,
                                             int64_t* d_recall_ids) {
    return shard_preassigned(h, p, nq, d_x, d_coarse_dis, d_probe, k, d_recall_dis, d_recall_ids, nullptr);
}

int gamma_hip_bound_combine(void* stream, float* d_acc, const float* d_in, int n, int take_max) {
    if (n < 0 || (n > 0 && (!d_acc || !d_in))) return GAMMA_HIP_EINVAL;
    gh::launch_bound_combine(static_cast<hipStream_t>(stream), d_acc, d_in, n, take_max);
    return hipGetLastError() == hipSuccess ? GAMMA_HIP_OK : GAMMA_HIP_EDEVICE;
}

int gamma_hip_ivfpq_search_shard_bounded(gamma_hip_index* h, const gamma_hip_search_params* p, int nq,
                                         const float* d_x, const float* d_coarse_dis, const int32_t* d_probe, int k,
                                         float* d_recall_dis, int64_t* d_recall_ids, float* d_bound,
                                         gamma_hip_bound_reduce_fn reduce, void* user) {
    ghi::BoundXchg bx;
    bx.d_bound = d_bound;
    bx.fn = reduce;
    bx.user = user;
    return shard_preassigned(h, p, nq, d_x, d_coarse_dis, d_probe, k, d_recall_dis, d_recall_ids, &bx);
}

static int merge_rerank_impl(gamma_hip_index* h, const gamma_hip_search_params* p, int nshards, int nq,
                             const float* d_x, int k, const float* d_all_dis, const int64_t* d_all_ids, const float* d_all_exact,
                             int q0, int nq_local, float* d_distances, int64_t* d_labels);
int gamma_hip_ivfpq_merge_rerank(gamma_hip_index* h, const gamma_hip_search_params* p, int nshards, int nq,
                                 const float* d_x, int k, const float* d_all_dis, const int64_t* d_all_ids,
                                 int q0, int nq_local, float* d_distances, int64_t* d_labels) {
    return merge_rerank_impl(h, p, nshards, nq, d_x, k, d_all_dis, d_all_ids, nullptr, q0, nq_local, d_distances, d_labels);
}
int gamma_hip_ivfpq_merge_rerank_exact(gamma_hip_index* h, const gamma_hip_search_params* p, int nshards, int nq,
                                       const float* d_x, int k, const float* d_all_dis, const int64_t* d_all_ids,
                                       const float* d_all_exact, int q0, int nq_local, float* d_distances, int64_t* d_labels) {
    if (!d_all_exact) return GAMMA_HIP_EINVAL;
    return merge_rerank_impl(h, p, nshards, nq, d_x, k, d_all_dis, d_all_ids, d_all_exact, q0, nq_local, d_distances, d_labels);
}
static int merge_rerank_impl(gamma_hip_index* h, const gamma_hip_search_params* p, int nshards, int nq,
                             const float* d_x, int k, const float* d_all_dis, const int64_t* d_all_ids, const float* d_all_exact,
                             int q0, int nq_local, float* d_distances, int64_t* d_labels) {
    if (!h) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    GH_TRY(replay_join(h));
    GH_TRY(ivfpq_check(h, p, nq, k));
    if (nshards <= 0 || q0 < 0 || nq_local < 0 || q0 + nq_local > nq) return fail(h, GAMMA_HIP_EINVAL, "bad shard/query range");
    if (k <= 0 || nq_local == 0) return GAMMA_HIP_OK;
    GH_CHECK(h, hipSetDevice(h->device));
    const bool l2 = p->metric == GAMMA_HIP_METRIC_L2;
    const int R = std::max(p->recall_num, k);
    if ((int64_t)nshards * R > (int64_t)1 << 24) return fail(h, GAMMA_HIP_EINVAL, "too many candidates");
    hipStream_t s = h->stream;
    GH_CHECK(h, h->w_cand_dis.ensure((size_t)nq_local * R * sizeof(float)));
    GH_CHECK(h, h->w_cand_ids.ensure((size_t)nq_local * R * sizeof(int64_t)));
    {
        StageScope t(h, GAMMA_HIP_STAGE_SELECT);
        static const bool no_merge_kernel = getenv("GAMMA_HIP_NO_MERGE_KERNEL") != nullptr;
        if (no_merge_kernel ||
            !gh::launch_merge_shards(s, l2, d_all_dis, d_all_ids, nshards, nq, R, q0, nq_local,
                                     h->w_cand_dis.as<float>(), h->w_cand_ids.as<int64_t>())) {
            // general shapes: transpose to [nq][W * R], select, translate positions to ids
            GH_CHECK(h, h->w_m_dis.ensure((size_t)nq * nshards * R * sizeof(float)));
            GH_CHECK(h, h->w_m_ids.ensure((size_t)nq * nshards * R * sizeof(int64_t)));
            GH_CHECK(h, h->w_cand_pos.ensure((size_t)nq_local * R * sizeof(int)));
            gh::launch_gather_shards(s, d_all_dis, d_all_ids, nshards, nq, R, h->w_m_dis.as<float>(),
                                     h->w_m_ids.as<int64_t>(), l2 ? INFINITY : -INFINITY);
            const int64_t stride = (int64_t)nshards * R;
            gh::launch_select_topk(s, l2, h->w_m_dis.as<float>() + (size_t)q0 * stride, stride, nullptr,
                                   (int)stride, (int)stride, nq_local, R, h->w_cand_dis.as<float>(),
                                   h->w_cand_pos.as<int>());
            gh::launch_take_ids(s, h->w_cand_pos.as<int>(), h->w_m_ids.as<int64_t>() + (size_t)q0 * stride, stride,
                                nq_local, R, h->w_cand_ids.as<int64_t>());
        }
        GH_CHECK(h, hipGetLastError());
    }
    // exact ties: the slice's queries whose result a tie can change are listed (gamma_hip_ivfpq_merge_flagged); the
    // caller gathers their candidate streams from the shards and has them replayed (gamma_hip_ivfpq_merge_replay)
    gamma_hip_search_params pp;
    GH_TRY(resolve_ties(h, p, &pp, p->nprobe <= gh::tie_replay_max_probes(), "exact_ties = 1 with nprobe > 1024"));
    p = &pp;
    const bool ties = tie_on(p);
    h->merge_flags = ties;
    h->merge_nql = nq_local;
    if (ties) {
        GH_CHECK(h, h->w_tcut.ensure((size_t)nq_local));
        GH_CHECK(h, h->w_tlist.ensure(((size_t)nq_local + 1) * sizeof(int)));
        GH_CHECK(h, hipMemsetAsync(h->w_tlist.p, 0, sizeof(int), s));
        gh::launch_flag_merge_cut(s, d_all_dis, nshards, nq, R, q0, nq_local, h->w_cand_dis.as<float>(), h->w_cand_ids.as<int64_t>(),
                                  h->w_tcut.as<uint8_t>(), h->merge_shard_flags);
    }
    h->merge_shard_flags = nullptr;   // one merge, with ties or without
    if (d_all_exact && p->has_rank) {
        // raw vectors sharded with their lists: the exact distance of every candidate travelled with it (computed by the shard
        // that holds the row, score window applied there).  compute_dis (gamma_index_ivfpq.cc:646-680) from those: the merged
        // candidates' distances are looked up in the shard tables, then the non-fused finish of stage B -- top-k of the row of
        // exact distances (equal distances keep the ADC order), output, tie flags.
        const float neutral = l2 ? 3.402823466e+38f : -3.402823466e+38f;
        StageScope t(h, GAMMA_HIP_STAGE_RERANK);
        GH_CHECK(h, h->w_exact.ensure((size_t)nq_local * R * sizeof(float)));
        GH_CHECK(h, h->w_selv.ensure((size_t)nq_local * k * sizeof(float)));
        GH_CHECK(h, h->w_selp.ensure((size_t)nq_local * k * sizeof(int)));
        gh::launch_lookup_exact(s, l2, d_all_dis, d_all_ids, d_all_exact, nshards, nq, R, q0, nq_local, h->w_cand_dis.as<float>(),
                                h->w_cand_ids.as<int64_t>(), h->w_exact.as<float>());
        gh::launch_select_topk(s, l2, h->w_exact.as<float>(), R, nullptr, R, R, nq_local, k, h->w_selv.as<float>(), h->w_selp.as<int>());
        gh::launch_finalize_topk(s, h->w_selv.as<float>(), h->w_selp.as<int>(), nq_local, k, h->w_cand_ids.as<int64_t>(), R, 0, neutral,
                                 d_distances, d_labels);
        if (ties) {
            gh::TieFlags tf;
            tf.cut = h->w_tcut.as<uint8_t>();
            tf.count = h->w_tlist.as<int>();
            tf.list = h->w_tlist.as<int>() + 1;
            tf.stats = h->d_tie_stats;
            GH_CHECK(h, h->w_textra.ensure((size_t)nq_local));
            GH_CHECK(h, hipMemsetAsync(h->w_textra.p, 0, (size_t)nq_local, s));
            gh::launch_flag_cut_ties(s, h->w_exact.as<float>(), R, nullptr, nq_local, k, h->w_selv.as<float>(), h->w_selp.as<int>(),
                                     nullptr, h->w_textra.as<uint8_t>(), R, 1);
            gh::launch_tie_list(s, tf.cut, h->w_textra.as<uint8_t>(), nq_local, tf.list, tf.count, tf.stats);
        }
        GH_CHECK(h, hipGetLastError());
        return GAMMA_HIP_OK;
    }
    return ivfpq_stage_b(h, p, nq_local, d_x + (size_t)q0 * h->d, R, k, h->w_cand_dis.as<float>(),
                         h->w_cand_ids.as<int64_t>(), d_distances, d_labels, nullptr, ties ? 2 : 0);
}

int gamma_hip_ivfpq_shard_exact(gamma_hip_index* h, const gamma_hip_search_params* p, int nq, const float* d_x,
                                const int64_t* d_ids, int R, float* d_exact) {
    if (!h || !p) return GAMMA_HIP_EINVAL;
    if (nq <= 0 || R <= 0) return GAMMA_HIP_OK;
    if (!d_x || !d_ids || !d_exact) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    GH_TRY(replay_join(h));
    if (!h->d_raw || h->raw_d != h->d) return fail(h, GAMMA_HIP_EINVAL, "no raw store");
    GH_CHECK(h, hipSetDevice(h->device));
    // (a dense store answers for every id; a sharded one for the vectors it holds, the sentinel elsewhere)
    gh::launch_rerank_dist(h->stream, p->metric == GAMMA_HIP_METRIC_L2, d_x, nq, h->d, h->d_raw, h->nraw, d_ids, R, p->min_score,
                           p->max_score, d_exact, h->raw_sparse ? h->d_raw_slot : nullptr, h->raw_sparse ? h->raw_slot_cap : 0);
    GH_CHECK(h, hipGetLastError());
    return GAMMA_HIP_OK;
}

int gamma_hip_ivfpq_shard_export_exact(gamma_hip_index* h, const gamma_hip_search_params* p, int nf, const float* d_xf,
                                       const float* d_vals, const int64_t* d_ids, const int32_t* d_off, int64_t stride,
                                       const float* d_bound_f, float* d_ex) {
    if (!h || !p) return GAMMA_HIP_EINVAL;
    if (nf <= 0) return GAMMA_HIP_OK;
    if (!d_xf || !d_vals || !d_ids || !d_off || !d_bound_f || !d_ex || stride < 1) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    if (!h->d_raw || h->raw_d != h->d || !h->raw_sparse) return fail(h, GAMMA_HIP_EINVAL, "export of exact distances: a sharded raw store (gamma_hip_raw_put)");
    GH_CHECK(h, hipSetDevice(h->device));
    gh::launch_export_exact(h->stream, p->metric == GAMMA_HIP_METRIC_L2, d_xf, nf, h->d, h->d_raw, h->d_raw_slot, h->raw_slot_cap,
                            d_vals, d_ids, stride, d_off, p->nprobe, d_bound_f, d_ex);
    GH_CHECK(h, hipGetLastError());
    return GAMMA_HIP_OK;
}

int gamma_hip_ivfpq_shard_cut_flags(gamma_hip_index* h, int nq, uint8_t* d_flags) {
    if (!h || nq < 0 || (nq > 0 && !d_flags)) return GAMMA_HIP_EINVAL;
    if (nq == 0) return GAMMA_HIP_OK;
    SearchLock lk(h);
    GH_CHECK(h, hipSetDevice(h->device));
    if (h->shard_cut_nq == nq) {
        GH_CHECK(h, hipMemcpyAsync(d_flags, h->shard_cut_chunked ? h->w_shard_cut.p : h->w_tcut.p, (size_t)nq,
                                   hipMemcpyDeviceToDevice, h->stream));
    } else {
        // no flags from the last shard search (exact ties off, or beyond the replay's range): "may have cut a tie"
        GH_CHECK(h, hipMemsetAsync(d_flags, 1, (size_t)nq, h->stream));
    }
    return GAMMA_HIP_OK;
}

int gamma_hip_ivfpq_merge_set_shard_flags(gamma_hip_index* h, const uint8_t* d_flags) {
    if (!h) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    h->merge_shard_flags = d_flags;
    return GAMMA_HIP_OK;
}

int gamma_hip_ivfpq_merge_flagged(gamma_hip_index* h, int* n_flagged, const int32_t** d_list) {
    if (!h || !n_flagged) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    *n_flagged = 0;
    if (d_list) *d_list = nullptr;
    if (!h->merge_flags) return GAMMA_HIP_OK;
    GH_CHECK(h, hipSetDevice(h->device));
    int n = 0;
    GH_CHECK(h, hipMemcpyAsync(&n, h->w_tlist.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    GH_CHECK(h, hipStreamSynchronize(h->stream));
    *n_flagged = std::min(n, h->merge_nql);
    if (d_list) *d_list = h->w_tlist.as<int32_t>() + 1;
    return GAMMA_HIP_OK;
}

int gamma_hip_ivfpq_max_list_len(gamma_hip_index* h) { return (h && h->ivf_init) ? h->max_list_len : 0; }

int gamma_hip_debug_heap_stream(gamma_hip_index* h, int op, int k, int n, const float* vals, float* arr_vals, int32_t* arr_ids,
                                float* sorted_vals, int32_t* sorted_ids) {
    if (!h || op < 0 || op > 4 || (op == 4 && n < 1) || k < 1 || k > gh::tie_small_max_k() || n < 0 || (n > 0 && !vals) || !arr_vals || !arr_ids ||
        !sorted_vals || !sorted_ids)
        return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    GH_CHECK(h, hipSetDevice(h->device));
    GH_CHECK(h, h->w_stage.ensure((size_t)std::max(n, 1) * sizeof(float) + (size_t)2 * k * sizeof(uint2) + 16));
    float* d_vals = h->w_stage.as<float>();
    uint2* d_arr = reinterpret_cast<uint2*>(h->w_stage.as<char>() + (((size_t)std::max(n, 1) * sizeof(float) + 15) & ~(size_t)15));
    uint2* d_sorted = d_arr + k;
    if (n > 0) GH_CHECK(h, hipMemcpyAsync(d_vals, vals, (size_t)n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    gh::launch_debug_heap_stream(h->stream, op, k, n, d_vals, d_arr, d_sorted);
    std::vector<uint2> out((size_t)2 * k);
    GH_CHECK(h, hipMemcpyAsync(out.data(), d_arr, (size_t)2 * k * sizeof(uint2), hipMemcpyDeviceToHost, h->stream));
    GH_CHECK(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < k; i++) {
        memcpy(&arr_vals[i], &out[i].x, 4);
        arr_ids[i] = (int32_t)out[i].y;
        memcpy(&sorted_vals[i], &out[k + i].x, 4);
        sorted_ids[i] = (int32_t)out[k + i].y;
    }
    return GAMMA_HIP_OK;
}

int gamma_hip_gather_rows(gamma_hip_index* h, const void* d_src, int row_words, const int32_t* d_list, int n, void* d_dst) {
    if (!h || row_words <= 0 || n < 0) return GAMMA_HIP_EINVAL;
    if (n == 0) return GAMMA_HIP_OK;
    if (!d_src || !d_list || !d_dst) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    GH_CHECK(h, hipSetDevice(h->device));
    gh::launch_gather_words(h->stream, d_src, d_list, n, row_words, d_dst);
    GH_CHECK(h, hipGetLastError());
    return GAMMA_HIP_OK;
}

int gamma_hip_ivfpq_shard_export_rows(gamma_hip_index* h, const gamma_hip_search_params* p, int nf, const int32_t* d_probe_f,
                                      int64_t* max_entries) {
    if (!h || !p || !max_entries) return GAMMA_HIP_EINVAL;
    *max_entries = 0;
    if (nf <= 0) return GAMMA_HIP_OK;
    if (!d_probe_f) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    GH_TRY(replay_join(h));
    if (!h->ivf_init || h->ivfflat) return fail(h, GAMMA_HIP_EINVAL, "ivfpq not initialised");
    GH_CHECK(h, hipSetDevice(h->device));
    GH_CHECK(h, h->w_qtotal.ensure(sizeof(int)));
    GH_CHECK(h, hipMemsetAsync(h->w_qtotal.p, 0, sizeof(int), h->stream));
    GH_CHECK(h, hipStreamWaitEvent(h->stream, h->ver_ev[h->cur_ver], 0));
    gh::launch_shard_export_rows(h->stream, d_probe_f, nf, p->nprobe, h->d_list_len, h->d_list_mask, h->nlist, h->w_qtotal.as<int>());
    int mx = 0;
    GH_CHECK(h, hipMemcpyAsync(&mx, h->w_qtotal.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    GH_CHECK(h, hipStreamSynchronize(h->stream));
    *max_entries = mx;
    return GAMMA_HIP_OK;
}

int gamma_hip_ivfpq_shard_export(gamma_hip_index* h, const gamma_hip_search_params* p, int nf, const float* d_xf,
                                 const float* d_cdis_f, const int32_t* d_probe_f, int64_t stride, float* d_vals, int64_t* d_ids,
                                 int32_t* d_off) {
    if (!h) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    GH_TRY(replay_join(h));
    GH_TRY(ivfpq_check(h, p, nf, 1));
    if (nf == 0) return GAMMA_HIP_OK;
    if (!d_xf || !d_cdis_f || !d_probe_f || !d_vals || !d_ids || !d_off) return fail(h, GAMMA_HIP_EINVAL, "null buffer");
    if (p->nprobe > gh::tie_replay_max_probes()) return fail(h, GAMMA_HIP_EINVAL, "nprobe beyond the replay's range");
    GH_CHECK(h, hipSetDevice(h->device));
    const int P = p->nprobe, R = std::max(p->recall_num, 1);
    if (stride < 1) return fail(h, GAMMA_HIP_EINVAL, "stride");
    gamma_hip_search_params pp = *p;   // the scan behind an export flags nothing: its top-R goes to scratch
    pp.exact_ties = -1;
    p = &pp;
    gh::FilterDesc filt;
    GH_TRY(build_filter(h, p, &filt));
    FiltCtx fc;
    GH_TRY(filt_ctx_single(h, filt, &fc));
    // the plain scan (every distance of the owned probed lists in the slab, scan order); its top-R goes to scratch
    const bool saved_bound = h->scan_bound;
    h->scan_bound = false;
    const int chunk = query_chunk(h, nf, P);
    int rc = GAMMA_HIP_OK;
    for (int f0 = 0; f0 < nf && rc == GAMMA_HIP_OK; f0 += chunk) {
        const int nc = std::min(chunk, nf - f0);
        rc = h->w_m_dis.ensure((size_t)nc * R * sizeof(float)) == hipSuccess &&
                     h->w_m_ids.ensure((size_t)nc * R * sizeof(int64_t)) == hipSuccess
                 ? GAMMA_HIP_OK
                 : fail(h, GAMMA_HIP_ENOMEM, "export scratch");
        if (rc == GAMMA_HIP_OK)
            rc = ivfpq_stage_a(h, p, fc.at(f0), nc, d_xf + (size_t)f0 * h->d, R, d_cdis_f + (size_t)f0 * P,
                               d_probe_f + (size_t)f0 * P, /*shard=*/true, h->w_m_dis.as<float>(), h->w_m_ids.as<int64_t>());
        if (rc == GAMMA_HIP_OK)
            gh::launch_shard_export(h->stream, d_probe_f + (size_t)f0 * P, nc, P, h->d_list_len, h->d_list_off, h->d_list_mask,
                                    h->nlist, h->d_ids, h->w_dist.as<float>(), h->tie.q_stride, stride,
                                    d_vals + (size_t)f0 * stride, d_ids + (size_t)f0 * stride, d_off + (size_t)f0 * (P + 1));
    }
    h->scan_bound = saved_bound;
    if (rc != GAMMA_HIP_OK) return rc;
    GH_CHECK(h, hipGetLastError());
    return GAMMA_HIP_OK;
}

static int merge_replay_impl(gamma_hip_index* h, const gamma_hip_search_params* p, int nshards, int nf, const float* d_x_slice,
                             int64_t stride, const float* d_vals_all, const int64_t* d_ids_all, const int32_t* d_off_all, const float* d_ex_all,
                             int k, const int32_t* d_list, float* d_distances, int64_t* d_labels);
int gamma_hip_ivfpq_merge_replay(gamma_hip_index* h, const gamma_hip_search_params* p, int nshards, int nf, const float* d_x_slice,
                                 int64_t stride, const float* d_vals_all, const int64_t* d_ids_all, const int32_t* d_off_all, int k,
                                 const int32_t* d_list, float* d_distances, int64_t* d_labels) {
    return merge_replay_impl(h, p, nshards, nf, d_x_slice, stride, d_vals_all, d_ids_all, d_off_all, nullptr, k, d_list, d_distances, d_labels);
}
int gamma_hip_ivfpq_merge_replay_exact(gamma_hip_index* h, const gamma_hip_search_params* p, int nshards, int nf, const float* d_x_slice,
                                       int64_t stride, const float* d_vals_all, const int64_t* d_ids_all, const int32_t* d_off_all,
                                       const float* d_ex_all, int k, const int32_t* d_list, float* d_distances, int64_t* d_labels) {
    if (!d_ex_all) return GAMMA_HIP_EINVAL;
    return merge_replay_impl(h, p, nshards, nf, d_x_slice, stride, d_vals_all, d_ids_all, d_off_all, d_ex_all, k, d_list, d_distances, d_labels);
}
static int merge_replay_impl(gamma_hip_index* h, const gamma_hip_search_params* p, int nshards, int nf, const float* d_x_slice,
                             int64_t stride, const float* d_vals_all, const int64_t* d_ids_all, const int32_t* d_off_all, const float* d_ex_all,
                             int k, const int32_t* d_list, float* d_distances, int64_t* d_labels) {
    if (!h) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    GH_TRY(ivfpq_check(h, p, nf, k));
    if (nf == 0 || k <= 0) return GAMMA_HIP_OK;
    if (!d_x_slice || !d_vals_all || !d_ids_all || !d_off_all || !d_list || !d_distances || !d_labels)
        return fail(h, GAMMA_HIP_EINVAL, "null buffer");
    if (!h->merge_flags) return fail(h, GAMMA_HIP_EINVAL, "no merge with tie flags before the replay");
    GH_CHECK(h, hipSetDevice(h->device));
    const bool l2 = p->metric == GAMMA_HIP_METRIC_L2;
    const int P = p->nprobe, R = std::max(p->recall_num, k);
    if (R > gh::tie_replay_max_k() || P > gh::tie_replay_max_probes()) return fail(h, GAMMA_HIP_EINVAL, "beyond the replay's range");
    if (p->has_rank && !d_ex_all && (!h->d_raw || h->raw_d != h->d || h->raw_sparse))
        return fail(h, GAMMA_HIP_EINVAL, "has_rank needs the raw store (or the exact distances exported with the streams)");
    hipStream_t s = h->stream;
    const int64_t mstride = ((int64_t)nshards * stride + 3) & ~(int64_t)3;   // a row assembled from every shard's export
    GH_CHECK(h, h->w_mr_vals.ensure((size_t)nf * mstride * sizeof(float)));
    GH_CHECK(h, h->w_mr_ids.ensure((size_t)nf * mstride * sizeof(int64_t)));
    GH_CHECK(h, h->w_mr_meta.ensure((size_t)nf * (P + 1) * sizeof(int32_t) + (size_t)nf * P * sizeof(int64_t) + 64));
    int64_t* m_base = h->w_mr_meta.as<int64_t>();
    int* count = reinterpret_cast<int*>(m_base + (size_t)nf * P);
    int32_t* m_off = count + 16;
    GH_CHECK(h, hipMemcpyAsync(count, &nf, sizeof(int), hipMemcpyHostToDevice, s));
    gh::launch_merge_streams(s, nshards, nf, P, stride, mstride, d_vals_all, d_ids_all, d_off_all, h->w_mr_vals.as<float>(),
                             h->w_mr_ids.as<int64_t>(), m_off, m_base, l2 ? INFINITY : -INFINITY);
    gh::TieReplayArgs a;
    if (d_ex_all && p->has_rank) {
        // the exported exact distances, assembled probe by probe exactly like the ADC values (same positions); the ids of this
        // second pass go to scratch.  A member of the recall_num-heap whose distance no shard exported is COUNTED and fails the call.
        GH_CHECK(h, h->w_fslab.ensure((size_t)nf * mstride * sizeof(float)));
        GH_CHECK(h, h->w_m_ids.ensure((size_t)nf * mstride * sizeof(int64_t)));
        GH_CHECK(h, h->w_lm_cnt.ensure(64));
        GH_CHECK(h, hipMemsetAsync(h->w_lm_cnt.p, 0, sizeof(int), s));
        gh::launch_merge_streams(s, nshards, nf, P, stride, mstride, d_ex_all, d_ids_all, d_off_all, h->w_fslab.as<float>(),
                                 h->w_m_ids.as<int64_t>(), m_off, m_base, __builtin_nanf(""));
        a.ex_slab = h->w_fslab.as<float>();
        a.ex_missing = h->w_lm_cnt.as<int>();
    }
    a.list = d_list;
    a.count = count;
    a.nq = nf;
    a.slab = h->w_mr_vals.as<float>();
    a.q_stride = mstride;
    a.pair_off = m_off;
    a.pair_base = m_base;
    a.ids = h->w_mr_ids.as<int64_t>();
    a.P = P;
    a.G = P;
    a.ready = nullptr;
    a.surv = nullptr;
    a.gcnt = nullptr;
    a.nsl = 0;
    a.slice_cap = 0;
    a.x = d_x_slice;
    a.d = h->d;
    a.raw = h->d_raw;
    a.nraw = h->nraw;
    a.R = R;
    a.k = k;
    a.has_rank = p->has_rank ? 1 : 0;
    a.min_score = p->min_score;
    a.max_score = p->max_score;
    a.neutral = l2 ? 3.402823466e+38f : -3.402823466e+38f;
    a.cand_dis = h->w_cand_dis.as<float>();     // the merged tables of the slice (gamma_hip_ivfpq_merge_rerank)
    a.cand_ids = h->w_cand_ids.as<int64_t>();
    a.distances = d_distances;
    a.labels = d_labels;
    a.compact_rows = 1;
    gh::launch_tie_replay(s, l2, a);
    GH_CHECK(h, hipGetLastError());
    if (a.ex_missing) {
        int missing = 0;
        GH_CHECK(h, hipMemcpyAsync(&missing, a.ex_missing, sizeof(int), hipMemcpyDeviceToHost, s));
        GH_CHECK(h, hipStreamSynchronize(s));
        if (missing) return fail(h, GAMMA_HIP_EDEVICE, "tie replay: a member of the recall_num-heap arrived without its exact distance");
    }
    return GAMMA_HIP_OK;
}

int gamma_hip_flat_search_device(gamma_hip_index* h, const gamma_hip_search_params* p, int nq,
                                 const float* d_x, int k, float* d_distances, int64_t* d_labels) {
    if (!h) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    GH_TRY(replay_join(h));
    return flat_search_device_locked(h, p, nq, d_x, k, d_distances, d_labels);
}

int gamma_hip_flat_search_device_wait(gamma_hip_index* h, const gamma_hip_search_params* p, int nq,
                                      const float* d_x, int k, float* d_distances, int64_t* d_labels) {
    if (!h) return GAMMA_HIP_EINVAL;
    hipEvent_t ev = nullptr;
    int* over = nullptr;
    {
        SearchLock lk(h);
        GH_CHECK(h, hipSetDevice(h->device));
        const unsigned slot = h->call_seq++ & 3u;
        if (!h->ev_call[slot]) GH_CHECK(h, hipEventCreateWithFlags(&h->ev_call[slot], hipEventDisableTiming));
        ev = h->ev_call[slot];
        if (!h->pin_flat_over) {
            GH_CHECK(h, hipHostMalloc((void**)&h->pin_flat_over, 4 * sizeof(int), hipHostMallocDefault));
            memset(h->pin_flat_over, 0, 4 * sizeof(int));
        }
        over = h->pin_flat_over + slot;
        *over = 0;
        h->defer_now = h->side2 != nullptr;
        h->flat_over_dst = over;
        const int rc = flat_search_device_locked(h, p, nq, d_x, k, d_distances, d_labels);
        h->defer_now = false;
        h->flat_over_dst = nullptr;
        if (rc != GAMMA_HIP_OK) return rc;
        // (the main stream carries the copy of the overflow word; the side stream, if a replay is pending, the last writes)
        if (h->replay_pending) {
            GH_CHECK(h, hipEventRecord(h->ev_rfork, h->stream));
            GH_CHECK(h, hipStreamWaitEvent(h->side2, h->ev_rfork, 0));
        }
        GH_CHECK(h, hipEventRecord(ev, h->replay_pending ? h->side2 : h->stream));
    }
    if (hipEventSynchronize(ev) != hipSuccess) return GAMMA_HIP_EDEVICE;
    if (*over) {   // a survivor list overflowed (adversarial data): the call again without a bound, the plain way
        SearchLock lk(h);
        h->flat_no_bound = true;
        int rc = replay_join(h);
        if (rc == GAMMA_HIP_OK) rc = flat_search_device_locked(h, p, nq, d_x, k, d_distances, d_labels);
        h->flat_no_bound = false;
        if (rc != GAMMA_HIP_OK) return rc;
        GH_CHECK(h, hipStreamSynchronize(h->stream));
    }
    return GAMMA_HIP_OK;
}

int gamma_hip_flat_search(gamma_hip_index* h, const gamma_hip_search_params* p, int nq, const float* x,
                          int k, float* distances, int64_t* labels) {
    if (!h) return GAMMA_HIP_EINVAL;
    // small unfiltered calls from concurrent client threads share device batches (see gamma_hip_ivfpq_search)
    if (h->combine && p && nq > 0 && nq <= COMB_MAX_NQ && k > 0 && x && distances && labels && h->raw_d > 0 &&
        !p->has_range && p->n_range == 0 && p->n_field == 0 && p->n_term == 0)
        return combined_search(h, p, nq, x, k, distances, labels, /*kind=*/1);
    return flat_search_host_locked(h, p, nq, x, k, distances, labels);
}


}  // extern "C"
