// libcfdenoise: everything that is decided or computed once per problem -- the fused cross-attention's work lists, the workspace of a
// problem (setup_problem), the timestep-only tables, the memory-side projections made once per run (prepare_static_memside) -- and the
// memory-side launches of a forward that still run per step (enqueue_memside).
#include "cfd_internal.hpp"

// ---- problem setup ----------------------------------------------------------------------------------
// Work list of the fused cross-attention kernel (xattn_fused.hpp).  A wave owns one tile of 16 queries of one batch row;
// the four waves of a workgroup share every K / V^T tile that passes through LDS, so
//  * rows are grouped by the instance of the LARGEST memory they attend to (the guidance batch: 5 of 7 chunks share the
//    unconditional audio memory; the other two chunks of an utterance share its own): a group's query tiles are dealt to
//    workgroups four at a time, and the long key stream is read once per workgroup whatever the rows' other memories are;
//  * per workgroup and memory, one segment per DISTINCT instance among its waves' rows (wave mask says who takes part);
//  * memories longer than one 32-key tile come first (online softmax; a flush of the accumulator between two of them);
//  * workgroups that read the same instance are placed on one XCD (block id % 8) next to each other so the stream is
//    fetched into that XCD's L2 once; big groups are dealt over all XCDs.
struct XaRow {      // one row of a work list
  int xrow;           // row of the residual stream (queries read from it; updated unless the list stores to xa_dedup)
  int inst[CFD_NMEM]; // memory instance per memory
  int aux;            // row of xa_dedup this row's tiles store to / add (-1: none)
  int one;            // instance of the one-key memory (Problem::xa_one), or -1
  int att;            // row of the attention-map blocks this row's tiles store to (Problem::att_fused), or -1
};

static void make_xattn_worklist(const Problem& p, const std::vector<XaRow>& rows, int mem_mask, std::vector<XaWg>& wgs, std::vector<XaSeg>& segs,
                                size_t& n_active, bool& has_flush) {
  const int L = p.L, nqt = (L + 15) / 16;
  wgs.clear(); segs.clear(); n_active = 0;
  // memory order: long (online) memories first, longest first; then the single-tile ones
  int order[CFD_NMEM], n_mem = 0, n_online = 0;
  for (int j = 0; j < CFD_NMEM; ++j)
    if ((mem_mask >> j) & 1) order[n_mem++] = j;
  if (n_mem == 0 || rows.empty()) return;
  std::stable_sort(order, order + n_mem, [&](int a, int b) { return p.Sp[a] > p.Sp[b]; });
  for (int oi = 0; oi < n_mem; ++oi) n_online += p.Sp[order[oi]] > XA_KEYS;
  const int jg = order[0];
  // groups of rows by instance of memory jg, in order of first appearance
  std::vector<int> inst_group(p.U[jg], -1);
  std::vector<std::vector<int>> groups;
  for (size_t r = 0; r < rows.size(); ++r) {
    int& g = inst_group[rows[r].inst[jg]];
    if (g < 0) { g = (int)groups.size(); groups.emplace_back(); }
    groups[g].push_back((int)r);
  }
  // Query tiles per workgroup.  A workgroup has room for XA_TILES = 4 (its K / V^T tiles then serve 64 queries), but a short list must
  // first of all FILL THE CHIP: at the product shape (L = 16: one tile per batch row) 32 utterances are 224 tiles = 56 workgroups for 256
  // CUs, each walking 14 segment steps because its four rows use different instances of the short memories (a pass per instance).  With
  // one tile per workgroup (six of the eight waves only request tile pieces) that is 224 workgroups of 9 steps: half the launch time.
  // Halve while the list stays at or below 128 workgroups.
  int tpw = XA_TILES;
  while (tpw > 1 && (rows.size() * (size_t)nqt + tpw - 1) / tpw <= 128) tpw /= 2;
  std::vector<std::vector<XaWg>> group_wgs(groups.size());
  for (size_t g = 0; g < groups.size(); ++g) {
    std::vector<std::pair<int, int>> tiles;   // (row of `rows`, first query)
    for (int r : groups[g])
      for (int t = 0; t < nqt; ++t) tiles.emplace_back(r, t * 16);
    for (size_t t0 = 0; t0 < tiles.size(); t0 += tpw) {
      XaWg w;
      memset(&w, 0, sizeof(w));
      int vr[XA_TILES];
      for (int k = 0; k < XA_TILES; ++k) {
        const bool on = k < tpw && t0 + k < tiles.size();
        vr[k] = on ? tiles[t0 + k].first : -1;
        w.row[k] = on ? rows[vr[k]].xrow : -1;
        w.aux[k] = on ? rows[vr[k]].aux : -1;
        w.one[k] = on ? rows[vr[k]].one : -1;
        w.att[k] = on ? rows[vr[k]].att : -1;
        w.q0[k] = on ? tiles[t0 + k].second : 0;
      }
      w.seg0 = (int)segs.size();
      int online_seen = 0;
      for (int oi = 0; oi < n_mem; ++oi) {
        const int j = order[oi];
        const bool online = p.Sp[j] > XA_KEYS;
        online_seen += online;
        int done = 0;
        size_t first_seg = segs.size();
        for (int k = 0; k < XA_TILES; ++k) {
          if (vr[k] < 0 || (done >> k) & 1) continue;
          XaSeg sg;
          sg.j = j; sg.u = rows[vr[k]].inst[j]; sg.wmask = 0; sg.flags = (online ? XA_ONLINE : 0) | (((p.xa_f16_mask >> j) & 1) ? XA_F16 : 0);
          for (int k2 = k; k2 < XA_TILES; ++k2)
            if (vr[k2] >= 0 && rows[vr[k2]].inst[j] == sg.u) sg.wmask |= 1 << k2;
          done |= sg.wmask;
          segs.push_back(sg);
        }
        // one accumulator: a finished online memory is flushed to x before the next online memory starts
        if (online && online_seen < n_online && segs.size() > first_seg) { segs.back().flags |= XA_FLUSH; has_flush = true; }
      }
      w.nseg = (int)segs.size() - w.seg0;
      w.n16 = 0;            // the segments with single-fp16 tiles: a prefix of the list (memories in descending length, XA_F16 <=> long enough)
      while (w.n16 < w.nseg && (segs[w.seg0 + w.n16].flags & XA_F16)) ++w.n16;
      for (int k = w.n16; k < w.nseg; ++k) segs[w.seg0 + k].flags &= ~XA_F16;   // (never: the order guarantees it; a flag behind the prefix would be read in the wrong format)
      group_wgs[g].push_back(w);
    }
  }
  // XCD placement: queue x holds the workgroups with block id % 8 == x, in dispatch order
  std::vector<std::vector<XaWg>> queue(8);
  std::vector<size_t> gorder(groups.size());
  for (size_t g = 0; g < groups.size(); ++g) gorder[g] = g;
  std::stable_sort(gorder.begin(), gorder.end(), [&](size_t a, size_t b) { return group_wgs[a].size() > group_wgs[b].size(); });
  auto shortest = [&]() { int q = 0; for (int x = 1; x < 8; ++x) if (queue[x].size() < queue[q].size()) q = x; return q; };
  for (size_t g : gorder) {
    if (group_wgs[g].size() > 32) { for (const XaWg& w : group_wgs[g]) queue[shortest()].push_back(w); }
    else { const int q = shortest(); for (const XaWg& w : group_wgs[g]) queue[q].push_back(w); }
  }
  size_t qlen = 0;
  for (int x = 0; x < 8; ++x) qlen = std::max(qlen, queue[x].size());
  XaWg idle;
  memset(&idle, 0, sizeof(idle));
  for (int k = 0; k < XA_TILES; ++k) { idle.row[k] = -1; idle.aux[k] = -1; idle.one[k] = -1; idle.att[k] = -1; }
  wgs.assign(qlen * 8, idle);
  for (int x = 0; x < 8; ++x) {
    for (size_t i = 0; i < queue[x].size(); ++i) wgs[i * 8 + x] = queue[x][i];
    n_active += queue[x].size();
  }
}

static int read_row_maps(Ctx* c, const cfd_memory mem[CFD_NMEM], std::vector<std::vector<int>>& hm) {
  const Problem& p = c->w->pb;
  hm.assign(CFD_NMEM, std::vector<int>(p.Be));
  for (int j = 0; j < CFD_NMEM; ++j) {
    if (mem[j].row_map) HIPCHK(hipMemcpy(hm[j].data(), mem[j].row_map, (size_t)p.Be * 4, hipMemcpyDeviceToHost));
    else for (int b = 0; b < p.Be; ++b) hm[j][b] = b;
    for (int b = 0; b < p.Be; ++b)
      if (hm[j][b] < 0 || hm[j][b] >= p.U[j]) return fail(CFD_E_ARG, "memory %s: row_map[%d] = %d outside [0, %d)", MEM_NAMES[j], b, hm[j][b], p.U[j]);
  }
  return CFD_OK;
}

static int upload_worklist(DBuf& dw, DBuf& ds, const std::vector<XaWg>& wgs, const std::vector<XaSeg>& segs) {
  CHK(dw.ensure(wgs.size() * sizeof(XaWg)));
  CHK(ds.ensure(segs.size() * sizeof(XaSeg)));
  HIPCHK(hipMemcpy(dw.p, wgs.data(), wgs.size() * sizeof(XaWg), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(ds.p, segs.data(), segs.size() * sizeof(XaSeg), hipMemcpyHostToDevice));
  return CFD_OK;
}

int build_xattn_worklist(Ctx* c, const cfd_memory mem[CFD_NMEM]) {
  Problem& p = c->w->pb;
  p.xa_nwg = 0; p.xa0_nwg_a = 0; p.xa0_nwg_b = 0; p.xa_flush = false;
  if (!c->fused_xattn) return CFD_OK;
  std::vector<std::vector<int>> hm;
  CHK(read_row_maps(c, mem, hm));
  // A memory of ONE key whose key no instance masks is added as a vector in the kernel's flush instead of walking a 32-key tile step
  // (xattn_fused.hpp, XAttnArgs::one_j): it gets no segments.  (With the key masked the reference's softmax is NaN: that stays a segment.)
  p.xa_one = -1;
  if (c->one_key && c->hoist_memside && p.tmode == 0) {
    for (int j = CFD_NMEM - 1; j >= 0 && p.xa_one < 0; --j) {
      if (p.S[j] != 1) continue;
      bool alive = true;
      if (mem[j].key_padding_mask) {
        std::vector<uint8_t> mk(p.U[j]);
        HIPCHK(hipMemcpy(mk.data(), mem[j].key_padding_mask, (size_t)p.U[j], hipMemcpyDeviceToHost));
        for (uint8_t v : mk) alive = alive && v == 0;
      }
      if (alive) p.xa_one = j;
    }
  }
  std::vector<XaRow> rows(p.Be);
  for (int b = 0; b < p.Be; ++b) {
    rows[b].xrow = b; rows[b].aux = -1; rows[b].one = p.xa_one >= 0 ? hm[p.xa_one][b] : -1;
    rows[b].att = (p.att_fused && b >= p.att_b0 && b < p.att_b0 + p.att_nb) ? b - p.att_b0 : -1;
    for (int j = 0; j < CFD_NMEM; ++j) rows[b].inst[j] = hm[j][b];
  }
  const int all_mems = ((1 << CFD_NMEM) - 1) & ~(p.xa_one >= 0 ? 1 << p.xa_one : 0);
  std::vector<XaWg> wgs;
  std::vector<XaSeg> segs;
  size_t n_active = 0;
  make_xattn_worklist(p, rows, all_mems, wgs, segs, n_active, p.xa_flush);
  if (wgs.empty() || segs.empty()) return CFD_OK;
  // A handful of workgroups cannot hide their serial walk over the key tiles (3 barriers and a fill round trip per 32 keys
  // with nothing else on the chip).  Round-2 measurements at the product shape, 1000 steps, since the memory-side projections
  // left the loop (the three-launch path still makes them per step): one utterance (2 workgroups) 1.342 s fused against
  // 1.318 s three-launch, four utterances (7 workgroups) 1.370 against 1.404, 16 (28 workgroups, one shard of the product-shape
  // benchmark) 470 against 465 steps/s.  Below 6 workgroups the three-launch path stays (CFD_FUSED_XATTN_MIN_WGS overrides;
  // the test suite sets 0 and runs its small cases through the fused kernel).
  // (counted in workgroups of four query tiles, as measured -- the list itself may deal fewer tiles per workgroup: make_xattn_worklist)
  if ((int)(((size_t)p.Be * ((p.L + 15) / 16) + XA_TILES - 1) / XA_TILES) < c->fused_xattn_min_wgs) return CFD_OK;
  (void)n_active;
  CHK(upload_worklist(c->w->xa_wgs, c->w->xa_segs, wgs, segs));
  p.xa_nwg = (int)wgs.size();
  return CFD_OK;
}

// Layer 0 of the sampling loop: the G guidance chunks of an utterance enter the first cross-attention with the SAME state (the
// replica-independent head, Problem::share_B), so the attention of that state against one memory instance is the same in every
// chunk that uses the instance.  For the longest memory (audio: 1 500 of the 1 573 keys at the benchmark shape) the 7 chunks of an
// utterance use 2 instances -- its own and the shared unconditional one -- so 2 evaluations replace 7:
//   list A  one row per distinct (utterance, instance of the longest memory): that memory only, result (incl. its rank-one
//           timestep term, without the bias) STORED to xa_dedup[aux]
//   list B  every row, the other memories, and x += ... + xa_dedup[aux(row)]
// Exact in real arithmetic; the summation order over the memories differs from the one-launch form (CFD_L0_DEDUP=0), so the G
// chunks of an utterance still leave layer 0 bit-identical where their memories are identical, but a run differs from the
// one-launch form by rounding (measured 2e-5 relative on final latents).
int build_xattn_layer0_lists(Ctx* c, const cfd_memory mem[CFD_NMEM]) {
  Problem& p = c->w->pb;
  p.xa0_nwg_a = p.xa0_nwg_b = 0;
  if (!c->l0_dedup || p.xa_nwg == 0 || p.share_B <= 0 || p.Be % p.share_B || p.Be == p.share_B) return CFD_OK;
  const int B = p.share_B, G = p.Be / B;
  std::vector<std::vector<int>> hm;
  CHK(read_row_maps(c, mem, hm));
  int jg = 0;
  for (int j = 1; j < CFD_NMEM; ++j)
    if (p.Sp[j] > p.Sp[jg]) jg = j;
  if (p.Sp[jg] < 256) return CFD_OK;   // nothing worth a second launch
  std::vector<XaRow> ra, rb(p.Be);
  std::vector<int> aux_of(p.Be, -1);
  for (int b = 0; b < B; ++b) {
    std::vector<std::pair<int, int>> seen;   // (instance, index in ra)
    for (int g = 0; g < G; ++g) {
      const int row = g * B + b, u = hm[jg][row];
      int idx = -1;
      for (auto& sn : seen)
        if (sn.first == u) idx = sn.second;
      if (idx < 0) {
        idx = (int)ra.size();
        XaRow r;
        r.xrow = row; r.aux = idx; r.one = -1;
        // (every chunk of the utterance enters layer 0 with the same state: the full-conditioning chunk's map against this instance is this row's)
        r.att = (p.att_fused && p.att_b0 + b < p.Be && hm[jg][p.att_b0 + b] == u) ? b : -1;
        for (int j = 0; j < CFD_NMEM; ++j) r.inst[j] = hm[j][row];
        ra.push_back(r);
        seen.emplace_back(u, idx);
      }
      aux_of[row] = idx;
    }
  }
  if (ra.size() * 2 > (size_t)p.Be) return CFD_OK;   // too little repetition
  for (int b = 0; b < p.Be; ++b) {
    rb[b].xrow = b; rb[b].aux = aux_of[b]; rb[b].one = p.xa_one >= 0 ? hm[p.xa_one][b] : -1;
    rb[b].att = (p.att_fused && b >= p.att_b0 && b < p.att_b0 + p.att_nb) ? b - p.att_b0 : -1;
    for (int j = 0; j < CFD_NMEM; ++j) rb[b].inst[j] = hm[j][b];
  }
  std::vector<XaWg> wa, wb;
  std::vector<XaSeg> sa, sb;
  size_t na = 0, nb = 0;
  make_xattn_worklist(p, ra, 1 << jg, wa, sa, na, p.xa_flush);
  make_xattn_worklist(p, rb, ((1 << CFD_NMEM) - 1) & ~(1 << jg) & ~(p.xa_one >= 0 ? 1 << p.xa_one : 0), wb, sb, nb, p.xa_flush);
  if (wa.empty() || wb.empty()) return CFD_OK;
  CHK(upload_worklist(c->w->xa0_wgs_a, c->w->xa0_segs_a, wa, sa));
  CHK(upload_worklist(c->w->xa0_wgs_b, c->w->xa0_segs_b, wb, sb));
  CHK(c->w->xa_dedup.ensure(ra.size() * (size_t)p.L * CFD_D * 4));
  p.xa0_nwg_a = (int)wa.size();
  p.xa0_nwg_b = (int)wb.size();
  return CFD_OK;
}

// Buffers and per-layer descriptors of the attention maps the fused cross-attention kernel keeps (Problem::att_fused; rows att_nb, set by the caller)
int setup_att_fused(Ctx* c) {
  Problem& pb = c->w->pb;
  XaAtt d;
  memset(&d, 0, sizeof(d));
  d.nb = pb.att_nb;
  for (int j = 0; j < CFD_NMEM; ++j) { d.off[j] = d.sp_tot; d.t0[j] = d.nt; d.sp_tot += pb.Sp[j]; d.nt += pb.Sp[j] / XA_KEYS; }
  const size_t rows = (size_t)pb.att_nb * pb.L;
  CHK(c->w->xa_att_raw.ensure(c->nl * rows * d.sp_tot * 4));
  CHK(c->w->xa_att_mc.ensure(c->nl * rows * d.nt * 4));
  CHK(c->w->xa_att_fin.ensure(c->nl * rows * CFD_NMEM * 2 * 4));
  CHK(c->w->xa_att_desc.ensure(c->nl * sizeof(XaAtt)));
  std::vector<XaAtt> desc(c->nl, d);
  for (int l = 0; l < c->nl; ++l) {
    desc[l].raw = c->w->xa_att_raw.as<float>() + (size_t)l * rows * d.sp_tot;
    desc[l].mc = c->w->xa_att_mc.as<float>() + (size_t)l * rows * d.nt;
    desc[l].fin = c->w->xa_att_fin.as<float>() + (size_t)l * rows * CFD_NMEM * 2;
  }
  HIPCHK(hipMemcpy(c->w->xa_att_desc.p, desc.data(), c->nl * sizeof(XaAtt), hipMemcpyHostToDevice));
  return CFD_OK;
}

int setup_problem(Ctx* c, int Be, int L, const cfd_memory mem[CFD_NMEM], float* const att[CFD_NMEM], int tmode, int T) {
  // (whatever this call is and however it ends, it may overwrite the memory-side buffers: the previous forward's projections are current
  //  only if cfd_forward says so again at its end)
  const bool mem_was_valid = c->w->fwd_mem_valid;
  c->w->fwd_mem_valid = false;
  if (!c->finalized) return fail(CFD_E_STATE, "weights not finalized");
  if (c->tsin_rows < 1) return fail(CFD_E_STATE, "timestep table not set");
  if (Be < 1 || L < 2) return fail(CFD_E_ARG, "bad batch / length");
  if (L % 2) return fail(CFD_E_SHAPE, "latent length %d is odd (reference: broadcasting error at position_encoding.py:160-161)", L);
  if (L / 2 > c->qpe_rows) return fail(CFD_E_SHAPE, "L/2 = %d exceeds the query PE buffer (%d rows)", L / 2, c->qpe_rows);
  if ((size_t)Be * 4 > c->w->iota.bytes) {   // identity row map (memories passed without de-duplication)
    CHK(c->w->iota.ensure((size_t)Be * 4));
    std::vector<int> id(Be);
    for (int i = 0; i < Be; ++i) id[i] = i;
    HIPCHK(hipMemcpy(c->w->iota.p, id.data(), (size_t)Be * 4, hipMemcpyHostToDevice));
  }
  Problem& p = c->w->pb;
  bool prev_same = mem_was_valid && c->w->fwd_wver == (unsigned long long)c->wver && c->w->fwd_Be == Be && tmode == 0;
  for (int j = 0; j < CFD_NMEM && prev_same; ++j)
    prev_same = c->w->fwd_U[j] == mem[j].U && c->w->fwd_S[j] == mem[j].S && c->w->fwd_mask[j] == (mem[j].key_padding_mask != nullptr) &&
                c->w->fwd_map[j] == (mem[j].row_map != nullptr);
  p.prev_same = prev_same;
  // ... and with the caller's promise that they ARE the same memories (cfd_forward_same_memories covers the row maps and masks), the work
  // lists and instance tables made from them -- several device-to-host reads per call -- are kept as well
  bool any_att_in = false;
  for (int j = 0; j < CFD_NMEM; ++j) any_att_in = any_att_in || (att && att[j]);
  const bool keep_lists = prev_same && c->hint_now && c->w->fwd_L == L && c->w->fwd_att == any_att_in;
  p.Be = Be; p.L = L; p.Lp = (L + 31) / 32 * 32; p.M = (long long)Be * L; p.tmode = tmode; p.T = T;
  p.share_B = 0;
  if (p.Lp > SM_MAX_CHUNKS * 512) return fail(CFD_E_SHAPE, "L = %d exceeds the in-register softmax limit (%d)", L, SM_MAX_CHUNKS * 512);
  int off = 0;
  for (int j = 0; j < CFD_NMEM; ++j) {
    const cfd_memory& m = mem[j];
    if (!m.data || m.U < 1 || m.S < 1) return fail(CFD_E_ARG, "memory %s: null/empty", MEM_NAMES[j]);
    if (!m.row_map && m.U != Be) return fail(CFD_E_ARG, "memory %s: U = %d != Be = %d without a row_map", MEM_NAMES[j], m.U, Be);
    if (tmode == 1 && m.row_map) return fail(CFD_E_ARG, "per-row timesteps need identity memory maps");
    if (m.S > c->mpe_rows)
      return fail(CFD_E_SHAPE, "memory %s has %d tokens, memory PE buffer has %d rows (reference: size mismatch at position_encoding.py:135)",
                  MEM_NAMES[j], m.S, c->mpe_rows);
    p.U[j] = m.U; p.S[j] = m.S; p.Sp[j] = (m.S + 31) / 32 * 32; p.off[j] = off; off += p.Sp[j];
    if (p.Sp[j] > SM_MAX_CHUNKS * 512) return fail(CFD_E_SHAPE, "memory %s: %d keys exceed the in-register softmax limit", MEM_NAMES[j], m.S);
    p.mem[j] = m.data; p.map[j] = m.row_map ? m.row_map : c->w->iota.as<int>(); p.mask[j] = m.key_padding_mask;
    p.att[j] = att ? att[j] : nullptr;
    p.att_slot[j] = 0;
    p.att_b0 = p.att_nb = 0;      // (a sampling run with an attention ring sets them after this call)
    p.att_fused = false;
    p.xa_opf = 0; p.xa_f16_mask = 0;
  }
  p.Sp_tot = off;
  // The run's operand policy (cfd_sample_begin): which memories are long enough for single-fp16 tiles.  The work lists flag their segments
  // (XA_F16); whether the run really takes the single-fp16 kernel instance is decided when everything else about it is known
  // (cfd_sample_begin, prepare_static_memside) -- the pair instance ignores the flag.
  p.xa_opf = c->want_opf;
#if !XA_ALL_OPF
  if (p.xa_opf) p.xa_opf = XA_V16 | XA_K16 | XA_P16 | XA_Q16;    // (the product builds ONE single-fp16 instance: all four bits; xattn_fused.hpp, XA_ALL_OPF)
#else
  if (p.xa_opf) p.xa_opf |= XA_V16 | XA_K16;                      // (developer builds: 3, 7, 11, 15 -- the tile formats always together)
#endif
  if (p.xa_opf)
    for (int j = 0; j < CFD_NMEM; ++j)
      if (p.Sp[j] >= XA_F16_MIN_KEYS) p.xa_f16_mask |= 1 << j;
  if (!p.xa_f16_mask) p.xa_opf = 0;
  {  // memories without a key-padding mask get an all-zero one, so the softmax kernel needs no null test
    size_t need = (size_t)Be * L;   // (the un-fused self-attention softmax indexes it per batch row)
    for (int j = 0; j < CFD_NMEM; ++j) need = std::max(need, (size_t)p.U[j] * p.S[j]);
    if (need > c->w->zero_mask.bytes) {
      CHK(c->w->zero_mask.ensure(need));
      HIPCHK(hipMemset(c->w->zero_mask.p, 0, need));
    }
    for (int j = 0; j < CFD_NMEM; ++j) {
      p.has_mask[j] = p.mask[j] != nullptr;
      if (!p.mask[j]) p.mask[j] = c->w->zero_mask.as<uint8_t>();
    }
  }
  p.jbig = -1; p.nruns = 0; p.nlong = 0; p.nshort = Be;
  if (c->use_runs && tmode == 0) {
    int jb = 0;
    for (int j = 1; j < CFD_NMEM; ++j)
      if (p.Sp[j] > p.Sp[jb]) jb = j;
    // worth it only for long latents and long memories (measured: 20.6 vs 21.2 ms at L=196 / 1500 keys, but
    // 3.60 vs 3.15 ms at L=16 / 161 keys, where the extra launches dominate)
    if (p.Sp[jb] >= 256 && L >= 64 && mem[jb].row_map) {
      std::vector<int> hmap(Be), lrows, srows;
      HIPCHK(hipMemcpy(hmap.data(), mem[jb].row_map, (size_t)Be * 4, hipMemcpyDeviceToHost));
      for (int b0 = 0; b0 < Be;) {
        int b1 = b0 + 1;
        while (b1 < Be && hmap[b1] == hmap[b0]) ++b1;
        if (b1 - b0 >= 4 && p.nruns < 8) {
          p.run_row0[p.nruns] = b0; p.run_len[p.nruns] = b1 - b0; p.run_u[p.nruns] = hmap[b0]; ++p.nruns;
          for (int b = b0; b < b1; ++b) lrows.push_back(b);
        } else {
          for (int b = b0; b < b1; ++b) srows.push_back(b);
        }
        b0 = b1;
      }
      if (p.nruns > 0) {
        p.jbig = jb; p.nlong = (int)lrows.size(); p.nshort = (int)srows.size();
        CHK(c->w->long_rows.ensure(lrows.size() * 4 + 16));
        CHK(c->w->short_rows.ensure(srows.size() * 4 + 16));
        HIPCHK(hipMemcpy(c->w->long_rows.p, lrows.data(), lrows.size() * 4, hipMemcpyHostToDevice));
        if (!srows.empty()) HIPCHK(hipMemcpy(c->w->short_rows.p, srows.data(), srows.size() * 4, hipMemcpyHostToDevice));
      }
    }
  }
  p.rt = c->rt_on && tmode == 0 && !g_cfd_naive_gemm && L <= RT_MAX_L && p.M <= c->rt_max_rows && p.Sp_tot <= RT_MAX_KEYS && c->hoist_memside;
  if (p.rt) {
    if (!keep_lists) p.rt_use_inst = Be <= RT_ARG_ROWS;
    for (int j = 0; j < CFD_NMEM && p.rt_use_inst && !keep_lists; ++j) {
      std::vector<int> hm(Be);
      if (mem[j].row_map) HIPCHK(hipMemcpy(hm.data(), mem[j].row_map, (size_t)Be * 4, hipMemcpyDeviceToHost));
      else for (int b = 0; b < Be; ++b) hm[b] = b;
      for (int b = 0; b < Be; ++b) {
        if (hm[b] < 0 || hm[b] >= p.U[j]) return fail(CFD_E_ARG, "memory %s: row_map[%d] = %d outside [0, %d)", MEM_NAMES[j], b, hm[b], p.U[j]);
        if (hm[b] > 255) p.rt_use_inst = 0;
        p.rt_inst[j][b] = (unsigned char)hm[b];
      }
    }
    CHK(c->w->rt_vt.ensure((size_t)Be * CFD_D * RT_MAX_L * 4));
    HIPCHK(hipMemset(c->w->rt_vt.p, 0, (size_t)Be * CFD_D * RT_MAX_L * 4));   // keys beyond L stay zero
  }
  // A forward that returns att_mats, beyond the row-tile path: the fused cross-attention kernel keeps every row's maps itself (its ATT
  // instance + att_fixup_kernel) instead of the three-launch path with its per-call memory-side projections.
  {
    bool any_att = false;
    for (int j = 0; j < CFD_NMEM; ++j) any_att = any_att || p.att[j];
    p.att_fused = any_att && !p.rt && tmode == 0 && c->att_fused && c->fused_xattn && c->hoist_memside && !g_cfd_naive_gemm;
    if (p.att_fused) { p.att_b0 = 0; p.att_nb = Be; }
  }
  if (!keep_lists) CHK(build_xattn_worklist(c, mem));
  if (p.att_fused && p.xa_nwg <= 0) { p.att_fused = false; p.att_nb = 0; }   // (a list too short for the fused kernel: three-launch path)
  if (p.att_fused && !keep_lists) CHK(setup_att_fused(c));
  const long long M = p.M;
  const int nl = c->nl;
  CHK(c->w->x.ensure((size_t)M * CFD_D * 4));
  CHK(c->w->h_sp.ensure((size_t)M * CFD_D * 4));
  CHK(c->w->qk_sp.ensure((size_t)M * 2 * CFD_D * 4));
  CHK(c->w->vts_sp.ensure((size_t)Be * CFD_D * ((L + 63) / 64 * 64) * 4));
  CHK(c->w->ssc.ensure((size_t)Be * CFD_NHEAD * L * p.Lp * 4));
  CHK(c->w->sp_sp.ensure((size_t)Be * CFD_NHEAD * L * p.Lp * 4));
  CHK(c->w->o_sp.ensure((size_t)M * CFD_D * 4));
  CHK(c->w->u_sp.ensure((size_t)M * CFD_FF * 4));
  CHK(c->w->sc.ensure((size_t)M * p.Sp_tot * 4));
  CHK(c->w->p_sp.ensure((size_t)M * p.Sp_tot * 4));
  CHK(c->w->eps.ensure((size_t)M * CFD_LAT * 4));
  CHK(c->w->sample_sp.ensure((size_t)M * CFD_LAT * 4));
  for (int j = 0; j < CFD_NMEM; ++j) {
    const size_t rows = (size_t)p.U[j] * p.Sp[j];
    CHK(c->w->n_sp[j].ensure(rows * CFD_D * 4));
    CHK(c->w->kall_sp[j].ensure(rows * nl * CFD_D * 4));
    CHK(c->w->cb[j].ensure(rows * (nl + 1) * 4));   // + one plane: the per-key scale of the fused cross-attention kernel
    CHK(c->w->vt_all[j].ensure(rows * nl * CFD_D * 4));
  }
  if (c->w->temb_tab.bytes < (size_t)T * CFD_D * 4 || c->w->ss_tab.bytes < (size_t)T * nl * 2 * 2 * CFD_D * 4) {
    c->w->tt_key.clear();          // (a table that is reallocated is an empty one: the timestep-only tables are rebuilt)
    c->w->tt_mem_mask = 0;
  }
  CHK(c->w->temb_tab.ensure((size_t)T * CFD_D * 4));
  CHK(c->w->h1_tab.ensure((size_t)T * CFD_D * 4));
  CHK(c->w->ss_tab.ensure((size_t)T * nl * 2 * 2 * CFD_D * 4));
  CHK(c->w->trows.ensure((size_t)T * 4));
  return CFD_OK;
}

// temb / TimeBlock modulation tables for the T timesteps in `trows_host` (embeddings.py:298-305,
// cross_attention.py:432-434).  temb depends only on t, so a sampling run computes all of its steps once.
int build_time_tables(Ctx* c, const int32_t* trows_host, int T, hipStream_t st) {
  for (int i = 0; i < T; ++i)
    if (trows_host[i] < 0 || trows_host[i] >= c->tsin_rows)
      return fail(CFD_E_ARG, "timestep %d outside the sinusoid table (0..%d)", trows_host[i], c->tsin_rows - 1);
  HIPCHK(hipMemcpyAsync(c->w->trows.p, trows_host, (size_t)T * 4, hipMemcpyHostToDevice, st));
  Work* w = c->w;
  const bool same = w->tt_wver == c->wver && (int)w->tt_key.size() == T && std::equal(w->tt_key.begin(), w->tt_key.end(), trows_host);
  if (same) return CFD_OK;          // temb_tab / ss_tab already hold these rows (and kbtab / vbtab may: tt_mem_mask)
  w->tt_key.assign(trows_host, trows_host + T);
  w->tt_wver = c->wver;
  w->tt_mem_mask = 0;
  c->setup_launches += 2 + 2 * c->nl;
  return enqueue_time_tables(c, T, st);
}

// the launches of build_time_tables: table rows from the timestep indices already in w->trows
int enqueue_time_tables(Ctx* c, int T, hipStream_t st) {
  const int ry = T < 64 ? T : 64;
  const float* W1 = rawp(c, "time_embedding.linear_1.weight");
  const float* b1 = rawp(c, "time_embedding.linear_1.bias");
  const float* W2 = rawp(c, "time_embedding.linear_2.weight");
  const float* b2 = rawp(c, "time_embedding.linear_2.bias");
  const int NE = c->nl * 2 * 2 * CFD_D;
  LAUNCH(CFD_PROF_OTHER, small_linear_kernel<>, dim3(CFD_D / 4, ry), dim3(256), st, c->tsin.as<float>(), c->w->trows.as<int>(),
         (long long)CFD_D, W1, b1, c->w->h1_tab.as<float>(), (long long)CFD_D, T, CFD_D, 0, 1);
  LAUNCH(CFD_PROF_OTHER, small_linear_kernel<>, dim3(CFD_D / 4, ry), dim3(256), st, c->w->h1_tab.as<float>(), (const int*)nullptr,
         (long long)CFD_D, W2, b2, c->w->temb_tab.as<float>(), (long long)CFD_D, T, CFD_D, 0, 0);
  // 18 emb_layers at once: rows (2l+tb)*1024 + n ; "post 2" adds 1 to the scale half
  // (post=2 tests n < 512 within each 1024 block -> handled by launching per time block)
  for (int tb = 0; tb < c->nl * 2; ++tb) {
    LAUNCH(CFD_PROF_OTHER, small_linear_kernel<>, dim3(2 * CFD_D / 4, ry), dim3(256), st, c->w->temb_tab.as<float>(), (const int*)nullptr,
           (long long)CFD_D, c->we_all.as<float>() + (size_t)tb * 2 * CFD_D * CFD_D, c->be_all.as<float>() + (size_t)tb * 2 * CFD_D,
           c->w->ss_tab.as<float>() + (size_t)tb * 2 * CFD_D, (long long)NE, T, 2 * CFD_D, 1, 2);
  }
  return CFD_OK;
}

// Once per cfd_forward / sampling run, after the time tables: the part of the memory-side work that does not depend on the
// timestep (see rows.hpp, mem_center_kernel, and xattn_fused.hpp).  Memories in `dynamic_mask` (contents rewritten between the
// iterations of a run: the dyadic rollout's partner projection) keep their per-step projections, and so does every memory when
// the fused cross-attention kernel is not the one that runs (att_mats wanted, small problems, per-row timesteps).
int prepare_static_memside(Ctx* c, hipStream_t st, int dynamic_mask, bool want_att, bool reuse) {
  Problem& p = c->w->pb;
  const int nl = c->nl;
  const long long ROWB = CFD_D * 4;
  p.static_mask = 0;
  for (int j = 0; j < CFD_NMEM; ++j) {   // scale plane = 1 unless mem_scale_all_kernel writes it
    const long long rows = (long long)p.U[j] * p.Sp[j];
    LAUNCH(CFD_PROF_OTHER, fill_f32_kernel<>, dim3((unsigned)((rows + 255) / 256)), dim3(256), st, c->w->cb[j].as<float>() + (size_t)nl * rows, rows, 1.0f);
  }
  if (c->w->zeros512.bytes == 0) {
    CHK(c->w->zeros512.ensure(CFD_D * 4));
    HIPCHK(hipMemsetAsync(c->w->zeros512.p, 0, CFD_D * 4, st));
  }
  if (dynamic_mask) p.rt = false;   // (a memory rewritten between iterations keeps its per-step projections: tile-kernel path)
  if (p.xa_one >= 0 && ((dynamic_mask >> p.xa_one) & 1)) {   // the one-key memory is rewritten between iterations: it needs its segments back
    cfd_memory mem[CFD_NMEM];
    memset(mem, 0, sizeof(mem));
    for (int j = 0; j < CFD_NMEM; ++j) {
      mem[j].data = p.mem[j]; mem[j].U = p.U[j]; mem[j].S = p.S[j]; mem[j].row_map = p.map[j];
      mem[j].key_padding_mask = p.has_mask[j] ? p.mask[j] : nullptr;
    }
    const int keep = c->one_key;
    c->one_key = 0;
    const int r = build_xattn_worklist(c, mem);
    c->one_key = keep;
    CHK(r);
  }
  const bool fused = p.rt || (c->fused_xattn && p.xa_nwg > 0 && !want_att && !g_cfd_naive_gemm);
  if (!fused || !c->hoist_memside || p.tmode != 0) { p.xa_opf = 0; return CFD_OK; }
  const int T = p.T;
  if (c->w->b_tab.bytes < (size_t)T * CFD_D * 4 || c->w->b_sp.bytes < (size_t)T * CFD_D * 4 || c->w->bsq.bytes < (size_t)T * 4)
    c->w->tt_mem_mask = 0;       // (a table that is reallocated is an empty one)
  CHK(c->w->b_tab.ensure((size_t)T * CFD_D * 4));
  CHK(c->w->b_sp.ensure((size_t)T * CFD_D * 4));
  CHK(c->w->bsq.ensure((size_t)T * 4));
  if (c->w->tt_mem_mask == 0) {
    c->w->tt_mem_mask = 0;
    LAUNCH(CFD_PROF_OTHER, temb_center_kernel<>, dim3((unsigned)((T + 3) / 4)), dim3(256), st, c->w->temb_tab.as<float>(), T, c->w->b_tab.as<float>(),
           c->w->b_sp.as<char>(), c->w->bsq.as<float>());
    c->setup_launches += 1;
  }
  for (int j = 0; j < CFD_NMEM; ++j) {
    if ((dynamic_mask >> j) & 1) continue;
    const int rows = p.U[j] * p.Sp[j];
    const int NK = nl * CFD_D + 32;
    CHK(c->w->ca[j].ensure((size_t)rows * nl * 4));
    CHK(c->w->asq[j].ensure((size_t)rows * 4));
    if (c->w->kbtab[j].bytes < (size_t)T * NK * 4 || c->w->vbtab[j].bytes < (size_t)T * nl * CFD_D * 4) c->w->tt_mem_mask &= ~(1 << j);
    CHK(c->w->kbtab[j].ensure((size_t)T * NK * 4));
    CHK(c->w->vbtab[j].ensure((size_t)T * nl * CFD_D * 4));
    const bool have_tb = (c->w->tt_mem_mask >> j) & 1;   // A_l b_t / VV_l b_t of this memory are already there for this timestep list
    MemCenterArgs ma{p.mem[j], p.U[j], p.S[j], p.Sp[j], rawp(c, "condition_embedding.weight") + (size_t)j * CFD_D, rawp(c, "mem_pos.pe"),
                     c->w->n_sp[j].as<char>(), c->w->asq[j].as<float>(), c->sat_mem()};
    // (`reuse`: the previous cfd_forward's memories again, cfd_forward_same_memories -- a_s, KA, ca and VA^T are in place)
    if (!reuse) LAUNCH(CFD_PROF_ROWS, mem_center_kernel<>, dim3((unsigned)((rows + 3) / 4)), dim3(256), st, ma);
    if (!reuse) {  // KA = A a_s for all layers, ca = c_l . a_s (-inf on dead keys)
      GemmArgs a = gemm_args();
      a.X[0] = c->wk_all_sp[j].as<char>(); a.ldx[0] = ROWB; a.I[0] = NK; a.Iclamp[0] = NK; a.kt[0] = CFD_D / 32;
      a.Y = c->w->n_sp[j].as<char>(); a.ldy = ROWB; a.J = rows; a.Jclamp = rows;
      a.super_i = 8; a.super_j = 8;
      EpiMemK e{c->w->kall_sp[j].as<char>(), (long long)rows, c->w->ca[j].as<float>(), nl * CFD_D, nl, p.mask[j], p.S[j], p.Sp[j], c->sat_mem()};
      CHK((run_gemm<MODE_PLAIN>(c, CFD_PROF_GEMM_MEM, a, e, 1, 1, st)));
    }
    if (!reuse) {  // VA^T
      GemmArgs a = gemm_args();
      a.X[0] = c->w->n_sp[j].as<char>(); a.ldx[0] = ROWB; a.I[0] = rows; a.Iclamp[0] = rows; a.kt[0] = CFD_D / 32;
      a.Y = c->wv_all_sp[j].as<char>(); a.ldy = ROWB; a.J = nl * CFD_D; a.Jclamp = nl * CFD_D;
      a.super_i = 8; a.super_j = 8;
      EpiMemV e{c->w->vt_all[j].as<char>(), p.Sp[j], p.U[j], c->sat_mem()};
      CHK((run_gemm<MODE_PLAIN>(c, CFD_PROF_GEMM_MEM, a, e, 1, 1, st)));
    }
    if (!have_tb) {  // kbtab[t][:] = [A_l b_t for all l | c_l . b_t]
      GemmArgs a = gemm_args();
      a.X[0] = c->wk_all_sp[j].as<char>(); a.ldx[0] = ROWB; a.I[0] = NK; a.Iclamp[0] = NK; a.kt[0] = CFD_D / 32;
      a.Y = c->w->b_sp.as<char>(); a.ldy = ROWB; a.J = T; a.Jclamp = T;
      EpiF32 e;
      memset(&e, 0, sizeof(e));
      e.out = c->w->kbtab[j].as<float>(); e.ldo = NK;
      CHK(run_gemm_plain_f32(c, CFD_PROF_GEMM_MEM, a, e, 1, 1, st));
    }
    if (!have_tb) {  // vbtab[t][:] = VV_l b_t for all l
      GemmArgs a = gemm_args();
      a.X[0] = c->wv_all_sp[j].as<char>(); a.ldx[0] = ROWB; a.I[0] = nl * CFD_D; a.Iclamp[0] = nl * CFD_D; a.kt[0] = CFD_D / 32;
      a.Y = c->w->b_sp.as<char>(); a.ldy = ROWB; a.J = T; a.Jclamp = T;
      EpiF32 e;
      memset(&e, 0, sizeof(e));
      e.out = c->w->vbtab[j].as<float>(); e.ldo = nl * CFD_D;
      CHK(run_gemm_plain_f32(c, CFD_PROF_GEMM_MEM, a, e, 1, 1, st));
      c->w->tt_mem_mask |= 1 << j;
      c->setup_launches += 2;
    }
    p.static_mask |= 1 << j;
  }
  if (p.xa_opf && (p.rt || p.static_mask != (1 << CFD_NMEM) - 1)) p.xa_opf = 0;   // (single-fp16 tiles: every memory static, tile kernels)
  if (p.xa_opf) {   // this run's operand policy: the key / value tiles of the fused cross-attention as single fp16, packed tile by tile
    for (int j = 0; j < CFD_NMEM; ++j) {
      if (!((p.xa_f16_mask >> j) & 1)) continue;   // (short memories keep pairs: xattn_fused.hpp, OPF)
      const long long tiles = (long long)nl * p.U[j] * (p.Sp[j] / XA_KEYS), chunks = tiles * 2048;
      if (p.xa_opf & XA_V16) {
        CHK(c->w->v16[j].ensure((size_t)tiles * 32768));
        LAUNCH(CFD_PROF_ROWS, xa_pack16_kernel<>, dim3((unsigned)((chunks + 255) / 256)), dim3(256), st, c->w->vt_all[j].as<char>(), c->w->v16[j].as<char>(), chunks, p.Sp[j], 0);
      }
      if (p.xa_opf & XA_K16) {
        CHK(c->w->k16[j].ensure((size_t)tiles * 32768));
        LAUNCH(CFD_PROF_ROWS, xa_pack16_kernel<>, dim3((unsigned)((chunks + 255) / 256)), dim3(256), st, c->w->kall_sp[j].as<char>(), c->w->k16[j].as<char>(), chunks, p.Sp[j], 1);
      }
    }
  }
  if (p.xa_one >= 0 && !p.rt) {   // the one-key memory's value rows as float32 vectors (xattn_fused.hpp, XAttnArgs::one_va).  Also with `reuse`:
                                  // the previous forward of these memories may have had another L or the row-tile path and never made them (one tiny launch)
    const int j = p.xa_one;
    const long long n = (long long)nl * p.U[j] * CFD_D;
    CHK(c->w->xa_one_va.ensure((size_t)n * 4));
    LAUNCH(CFD_PROF_ROWS, one_key_va_kernel<>, dim3((unsigned)((n + 255) / 256)), dim3(256), st, c->w->vt_all[j].as<char>(), n, p.Sp[j], c->w->xa_one_va.as<float>());
  }
  if (p.rt) {   // per-key scale and key bias of every step of the run (the tile-kernel path makes one step's per iteration: mem_scale_all_kernel)
    for (int j = 0; j < CFD_NMEM; ++j) {
      const long long rows = (long long)p.U[j] * p.Sp[j];
      CHK(c->w->rt_cbt[j].ensure((size_t)T * (nl + 1) * rows * 4));
      MemScaleTabArgs a{c->w->n_sp[j].as<char>(), c->w->asq[j].as<float>(), rows, c->w->b_tab.as<float>(), c->w->bsq.as<float>(), c->w->ca[j].as<float>(),
                        c->w->kbtab[j].as<float>() + (size_t)nl * CFD_D, (long long)(nl * CFD_D + 32), nl, c->w->rt_cbt[j].as<float>()};
      LAUNCH(CFD_PROF_ROWS, mem_scale_table_kernel<>, dim3((unsigned)((rows + 3) / 4), (unsigned)T), dim3(256), st, a);
    }
  }
  return CFD_OK;
}

// memory-side work of one forward: shared by every row chunk
int enqueue_memside(Ctx* c, hipStream_t st) {
  const Problem& p = c->w->pb;
  if (p.rt) return CFD_OK;   // every memory is static and its per-step scalars are tabulated (prepare_static_memside)
  const int nl = c->nl;
  const int* dstep = p.tmode ? c->w->d_step.as<int>() + 1 : c->w->d_step.as<int>();
  const long long ROWB = CFD_D * 4;
  const dim3 blk(256);
  // memories whose projections were made once for the run: this step's per-key scale and key bias
  {
    MemScaleAllArgs g;
    memset(&g, 0, sizeof(g));
    int nwg = 0;
    for (int j = 0; j < CFD_NMEM; ++j) {
      if (!((p.static_mask >> j) & 1)) continue;
      const long long rows = (long long)p.U[j] * p.Sp[j];
      const int NK = nl * CFD_D + 32;
      g.m[g.n] = MemScaleArgs{c->w->n_sp[j].as<char>(), c->w->asq[j].as<float>(), rows, c->w->b_tab.as<float>(), c->w->bsq.as<float>(), c->w->ca[j].as<float>(),
                              c->w->kbtab[j].as<float>() + (size_t)nl * CFD_D, (long long)NK, dstep, nl, c->w->cb[j].as<float>() + (size_t)nl * rows,
                              c->w->cb[j].as<float>()};
      g.first[g.n] = nwg;
      nwg += (int)((rows + 3) / 4);
      ++g.n;
    }
    g.first[g.n] = nwg;
    if (g.n > 0) LAUNCH(CFD_PROF_ROWS, mem_scale_all_kernel<>, dim3((unsigned)nwg), blk, st, g);
  }
  // 2. memories: + temb + condition id + PE, normalise           (denoiser.py:223-261,332-353)
  for (int j = 0; j < CFD_NMEM; ++j) {
    if ((p.static_mask >> j) & 1) continue;
    MemPrepArgs a{p.mem[j], p.U[j], p.S[j], p.Sp[j], c->w->temb_tab.as<float>(), dstep, p.tmode,
                  rawp(c, "condition_embedding.weight") + (size_t)j * CFD_D, rawp(c, "mem_pos.pe"), c->w->n_sp[j].as<char>()};
    const long long rows = (long long)p.U[j] * p.Sp[j];
    LAUNCH(CFD_PROF_ROWS, mem_prep_kernel<>, dim3((unsigned)((rows + 3) / 4)), blk, st, a);
  }
  // 3. memory-side projections for ALL layers at once: folded keys (+ key bias) and folded values^T
  for (int j = 0; j < CFD_NMEM; ++j) {
    if ((p.static_mask >> j) & 1) continue;
    const int rows = p.U[j] * p.Sp[j];
    c->memside_in_forward = true;   // these epilogues count into the handle's census: whoever waits for this stream next reads it
    {
      GemmArgs a = gemm_args();
      a.X[0] = c->wk_all_sp[j].as<char>(); a.ldx[0] = ROWB; a.I[0] = nl * CFD_D + 32; a.Iclamp[0] = nl * CFD_D + 32; a.kt[0] = CFD_D / 32;
      a.Y = c->w->n_sp[j].as<char>(); a.ldy = ROWB; a.J = rows; a.Jclamp = rows;
      a.super_i = 8; a.super_j = 8;
      EpiMemK e{c->w->kall_sp[j].as<char>(), (long long)rows, c->w->cb[j].as<float>(), nl * CFD_D, nl, p.mask[j], p.S[j], p.Sp[j], c->sat_mem()};
      CHK((run_gemm<MODE_PLAIN>(c, CFD_PROF_GEMM_MEM, a, e, 1, 1, st)));
    }
    {
      GemmArgs a = gemm_args();
      a.X[0] = c->w->n_sp[j].as<char>(); a.ldx[0] = ROWB; a.I[0] = rows; a.Iclamp[0] = rows; a.kt[0] = CFD_D / 32;
      a.Y = c->wv_all_sp[j].as<char>(); a.ldy = ROWB; a.J = nl * CFD_D; a.Jclamp = nl * CFD_D;
      a.super_i = 8; a.super_j = 8;
      EpiMemV e{c->w->vt_all[j].as<char>(), p.Sp[j], p.U[j], c->sat_mem()};
      CHK((run_gemm<MODE_PLAIN>(c, CFD_PROF_GEMM_MEM, a, e, 1, 1, st)));
    }
  }

  return CFD_OK;
}

