// chm_fused.h -- k_marg_fused: ONE kernel per (event, draw) for the standard marginalized configuration (likelihood.py:160-205, 266-281 with
// pop_wrapper.py:67-90): det -> src conversion and weights of the event's samples, event statistics, the 32 per-pixel histograms, the per-z
// factors on the support of the event's KDE, prefix sums + KDE + integrand per pixel, the pixel sum and log L_i -- z and w never leave the CU.
//
// What it replaces (chm_kernels.h): k_samples_fast -> (z, w) workspace in HBM -> k_event_stats / k_zfactors -> k_kde_marg_sub2 -> k_marg_fixup.
// The arithmetic per sample / node / grid point is the SAME code (z_from_lut_x2, p_m1m2_fused, bin_index_r, zfactor_point, kde_sub_item<PRE>);
// what differs is the order of the floating-point sums that carry no order in the reference either (a bin's weights, likelihood.py:183 /
// math.py:42 `.at[].add`; the event's sum w): results agree with the separate kernels to rounding (~1e-15 per event), not bit for bit.
//
// Why one pass over the samples is enough: the histogram of pixel p needs lo = min z (event) and hi_p = max z (pixel) BEFORE the first sample is
// binned (math.py:36-41).  z_from_dGW is a monotone interpolation of a sorted table (the draw's `fits` flag), so min z = z(min dL) and
// max z = z(max dL): the extreme distances of every event and pixel are draw-independent and found once at upload (dl_lo, dl_hi, pix_dlmax).
// Draws whose table is not sorted and events with a non-finite or non-positive distance take the exact route: a pre-pass writes z to the
// workspace and reduces min / max from it.
//
// Block = NW waves, one (event, draw).  LDS (doubles unless noted):
//   H[rows][B+1]   per-pixel histograms; row p becomes the prefix array P0 of pixel p in place (kde_sub_item<PRE>)
//   misc           event statistics, per-wave partial sums, per-pixel {hi - lo, 1/(hi - lo)}, hi, integral, rounding bound
//   overlay        sample pass: exp table | windows of m_grid, cdf_m2 | the event's slice of the node records | of the direct-index table (u16) |
//                  NW - 1 boundary rows;  pixel pass: -2 P1 | P2 of the two pixels each wave has in hand
// Determinism: wave w owns the contiguous tiles [w NT/NW, (w+1) NT/NW) of the (pixel-sorted) samples.  A pixel whose segment began in an
// earlier wave's range is that wave's `boundary pixel`: its weights go to the wave's own boundary row, and the rows are added to the pixel's
// histogram in wave order after the pass -- every bin is summed in one fixed order, run after run.
#pragma once

struct FusedDesc {
  const unsigned char* pix_id;    // (E, NT*128) local pixel of every (pixel-sorted) sample, 255: in no pixel / padding
  const double* pix_dlmax;        // (E, P) largest distance among the pixel's samples (NaN: no sample)
  const unsigned char* ev_plain;  // (E) 1: every distance of the event is finite and positive
  const double2* lm1;             // (E) log of the smallest / largest finite positive m1det of the event (NaN: none)
  int cap_rec, cap_keys;          // LDS rows / entries reserved for an event's slice of the node records / of the direct-index table
  int overlay_doubles;            // size of the overlay region
  int cap_m;                      // LDS entries reserved for the window of m_grid / cdf_m2 an event's source-frame masses can reach
  double tol;                     // dense redo when the summed rounding bound exceeds tol L_i (k_marg_fixup's criterion)
  int* redo_count;                // diagnostics: number of (event, draw) pairs that took the dense redo (may be NULL)
};

#ifndef CHM_FUSED_NPV
#define CHM_FUSED_NPV 16
#endif

template <int MASS, int NW, int BINS, bool NT>
__global__ void __launch_bounds__(64 * NW, NW >= 16 ? 4 : 2)
k_marg_fused(LikeDev L, SampFast F, FusedDesc D, const DevParams* params, const double* zt_all, const double* It_all, const double* dLt_all,
             const double* mg_all, const double* cdf_all, const double* rec_all, int TcMax, int TmMax) {
#pragma clang fp contract(fast)                  // sums of products may fuse; z (z_from_lut_x2 / jnp_interp) and the bin index (bin_index_r) are formed in bodies with contraction off
  extern __shared__ double lds[];
  constexpr int NT_ = 64 * NW;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int b = blockIdx.x % L.nb, e = L.e_off + blockIdx.x / L.nb;
  PH_INIT;
  DevParams P = params[b];
  mass_params_to_vgpr<MASS, CHM_FUSED_NPV>(P);
  const int B = BINS > 0 ? BINS : L.num_bins, HS = B + 1, Pn = L.P, S = L.S, Z = L.Z;
  const int Hrows = Pn + (Pn & 1);
  const int Tc = P.Tc, Tm = P.Tm;
  // ---- LDS carve-up
  double* const H = lds;
  double* const esv = H + (size_t)Hrows * HS;               // [16] event statistics (NEVSTAT) + [12] z_ref, [13] zmax, [14] flags
  double* const red = esv + 16;                             // [NW][4]
  double* const pxp = red + NW * 4;                         // [Pn][2]  hi - lo, 1/(hi - lo)
  double* const pxh = pxp + 2 * (size_t)Hrows;              // [Pn]     hi
  double* const lik = pxh + Hrows;                          // [Pn]
  double* const err = lik + Hrows;                          // [Pn]
  int* const bnd = reinterpret_cast<int*>(err + Hrows);     // [NW] boundary pixel of every wave (-1: none), [NW] spare
  double* const ov = reinterpret_cast<double*>(bnd + 2 * NW);
  double* const etab = ov;
  double* const mg = etab + CHM_EXPTAB_N; double* const cdf = mg + D.cap_m;
  double* const rec = cdf + D.cap_m;
  unsigned short* const luts = reinterpret_cast<unsigned short*>(rec + 4 * (size_t)D.cap_rec);
  double* const Hb = rec + 4 * (size_t)D.cap_rec + (D.cap_keys + 3) / 4;      // [NW - 1][HS] boundary rows
  const ExpTab ex = { etab };
  const double* g_zt = zt_all + (size_t)b * TcMax;
  const double* g_It = It_all + (size_t)b * TcMax;
  const double* g_dLt = dLt_all + (size_t)b * TcMax;
  const size_t eo = (size_t)e * S;
  const size_t so = ((size_t)b * L.E + e) * S;
  const size_t be = (size_t)b * L.E + e;
  const double* zg = L.z_grids + (size_t)e * Z;
  const int* seg = L.seg_off + (size_t)e * (Pn + 1);
  const int NTe = F.NT;
  const unsigned char* pid = D.pix_id + (size_t)e * NTe * SF_TILE;
  const double2* tbase = reinterpret_cast<const double2*>(F.tiles + (size_t)e * NTe * 6 * SF_TILE) + lane;

  // ---- the event's slice of the draw's direct-index table and node records
  const int* info = F.lut.info + (size_t)b * 4;
  const int i_lo = info[0], ns = info[1], lmax = info[2];
  const int key0s = F.lut.key0, nks = F.lut.nk;
  const unsigned short* gl = F.lut.lut + (size_t)b * (nks + 1);
  const double xlo = L.dl_lo[e], xhi = L.dl_hi[e];
  const int klo = lut_key(xlo, key0s), khi = lut_key(xhi, key0s);
  bool fast = info[3] != 0 && D.ev_plain[e] != 0 && (unsigned)klo < (unsigned)nks && (unsigned)khi < (unsigned)nks && khi >= klo;
  int rlo = 0, rhi = 0, ns_e = 1, nk_e = 1;
  if (fast) {
    const int glo = gl[klo], ghi = gl[khi];
    rlo = glo > 0 ? glo - 1 : 0;
    // an event whose smallest distance lies beyond the table's last node still needs the record of the LAST interval (z clamps to z_last through it,
    // z_from_lut_x2): without this line its slice began at node Tc - 1 and the clamp read the record in front of the slice (scripts/fuzz_parity.py, round 4)
    rlo = rlo > Tc - 2 ? (Tc >= 2 ? Tc - 2 : 0) : rlo;
    rlo = rlo < i_lo ? i_lo : rlo;
    rhi = ghi + lmax; rhi = rhi > i_lo + ns - 1 ? i_lo + ns - 1 : rhi;
    ns_e = rhi - rlo + 1; nk_e = khi - klo + 1;
    fast = ns_e >= 1 && ns_e <= D.cap_rec && nk_e + 1 <= D.cap_keys;
  }
  const int key0e = key0s + klo;
  const double x_last = g_dLt[Tc - 1], z_last = g_zt[Tc - 1];
  // an event with a distance beyond the table's last node (z clamps at z_max) takes the exact route: the slice arithmetic above is laid out for
  // distances inside the table, and two all-beyond events of the round-4 fuzz run came out with run-to-run garbage on the fast route
  fast = fast && xhi <= x_last;
  // window of the mass grid the event's source-frame primary masses can reach: log m1 = log m1det - log(1 + z) with log m1det in the event's
  // [lm1.x, lm1.y] and z between the first and one past the last node record of its slice (their log(1 + z) are in the records); 4 entries of
  // margin on either side (rounding of the position, the stepping loops of p_m1m2_fused).  A window that does not fit: tables read from L2.
  const double* gm = mg_all + (size_t)b * TmMax;
  const double* gc = cdf_all + (size_t)b * TmMax;
  int mlo = 0, mlen = Tm;
  bool mass_lds = Tm <= D.cap_m;
  if (fast && !mass_lds) {
    const double* recb = rec_all + (size_t)b * TcMax * 4;
    const double lzA = recb[4 * rlo + 3], lzB = recb[4 * (rhi + 1 < Tc ? rhi + 1 : Tc - 1) + 3];
    const double2 lmb = D.lm1[e];
    auto pos = [&](double lm) { const double tt = (lm - P.lmg0) * P.inv_dlmg; const int i = (tt >= 0.) ? (tt < (double)Tm ? (int)tt + 1 : Tm - 1) : 1; return i < 1 ? 1 : (i > Tm - 1 ? Tm - 1 : i); };
    int ia = pos(lmb.x - lzB) - 5, ib = pos(lmb.y - lzA) + 4;
    ia = ia < 0 ? 0 : ia; ib = ib > Tm - 1 ? Tm - 1 : ib;
    if (lmb.x == lmb.x && lmb.y == lmb.y && lzA == lzA && lzB == lzB && ib - ia + 1 <= D.cap_m) { mlo = ia; mlen = ib - ia + 1; mass_lds = true; }
  }
  const TabSlice mgs = { mg - mlo, P.mg_first, P.mg_last }, cdfs = { cdf - mlo, gc[0], P.cdf_last };

  auto stage_tables_ev = [&]() {
    for (int i = t; i < CHM_EXPTAB_N; i += NT_) etab[i] = exp_table_entry(i);
    if (mass_lds) for (int i = t; i < mlen; i += NT_) { mg[i] = gm[mlo + i]; cdf[i] = gc[mlo + i]; }
    if (fast) {
      const double2* gr = reinterpret_cast<const double2*>(rec_all + ((size_t)b * TcMax + rlo) * 4);
      double2* lr = reinterpret_cast<double2*>(rec);
      for (int i = t; i < 2 * ns_e; i += NT_) lr[i] = gr[i];
      for (int i = t; i <= nk_e; i += NT_) luts[i] = gl[klo + i];
    }
  };
  stage_tables_ev();
  for (int i = t; i < Hrows * HS; i += NT_) H[i] = 0.;
  for (int i = t; i < (NW - 1) * HS; i += NT_) Hb[i] = 0.;
  // this wave's tiles and its boundary pixel (the pixel its first sample belongs to, if that pixel's segment began before the wave's range)
  const int tb0 = (int)(((long long)w * NTe) / NW), tb1 = (int)(((long long)(w + 1) * NTe) / NW);
  int pfirst = -1;
  if (w > 0 && tb0 < tb1 && tb0 * SF_TILE < S) {
    const int q = pid[tb0 * SF_TILE];
    if (q < Pn && seg[q] < tb0 * SF_TILE) pfirst = q;
  }
  if (lane == 0) bnd[w] = pfirst;
  __syncthreads();
  PH(0);

  // ---- lo = min z, zmax = max z of the event, hi_p = max z of every pixel
  double lo, zmx;
  if (fast) {
    if (t <= Pn) {
      const double xa = t < Pn ? D.pix_dlmax[(size_t)e * Pn + t] : xhi;
      double za, zb, d0, d1, d2, d3; bool bad;
      z_from_lut_x2(xa == xa ? xa : xlo, xlo, rec, luts, key0e, nk_e, rlo, ns_e, lmax, Tc, x_last, z_last, za, zb, d0, d1, d2, d3, bad);
      if (t < Pn) { const double dhl = za - zb; pxh[t] = za; pxp[2 * t] = dhl; pxp[2 * t + 1] = 1. / dhl; }     // a pixel without samples: hi = lo (degenerate: NaN below, as the reference's 0/0)
      else { esv[0] = zb; esv[13] = za; }
    }
  } else {
    // exact route: z of every sample by the general search on the global tables -> workspace; min / max reduced from there
    double mn = __builtin_inf(), mxx = -__builtin_inf();
    for (int s = 2 * t; s < S; s += 2 * NT_) {
      const double2 dl = *(tbase - lane + (size_t)(s / SF_TILE) * (6 * SF_TILE / 2) + ((s % SF_TILE) >> 1));
      double za, zb;
      z_from_dGW_x2(P, dl.x, dl.y, g_dLt, g_zt, za, zb);
      L.ws_z[so + s] = za; mn = nanmin2(mn, za); mxx = nanmax2(mxx, za);
      if (s + 1 < S) { L.ws_z[so + s + 1] = zb; mn = nanmin2(mn, zb); mxx = nanmax2(mxx, zb); }
    }
    mn = block_reduce<RED_MIN>(mn, red);
    mxx = block_reduce<RED_MAX>(mxx, red);                  // (block_reduce's barriers also order the workspace stores before the reads below)
    __threadfence_block();
    if (t == 0) { esv[0] = mn; esv[13] = mxx; }
    for (int p = w; p < Pn; p += NW) {                      // hi = max(where(mask, z, min z)) (likelihood.py:180): NaN-ignoring from lo, NaN when lo is
      double hi = mn;
      for (int s = seg[p] + lane; s < seg[p + 1]; s += 64) hi = vmax_f64(hi, L.ws_z[so + s]);
      hi = wave_max_dpp(hi);
      if (mn != mn) hi = mn;
      if (lane == 0) { const double dhl = hi - mn; pxh[p] = hi; pxp[2 * p] = dhl; pxp[2 * p + 1] = 1. / dhl; }
    }
  }
  __syncthreads();
  lo = esv[0]; zmx = esv[13];
  PH(1);

  // ---- sample pass: z, w, statistics, histogram scatter
  // reference point of the shifted sums (k_samples_fast's): the table node next to the event's smallest distance
  double z_ref;
  if (fast) { int q = (int)luts[0] - rlo; q = q < 0 ? 0 : (q > ns_e - 1 ? ns_e - 1 : q); z_ref = rec[4 * q + 1]; }
  else { const int c_lo = (P.dl_sorted != 0. && xlo == xlo) ? searchsorted_right(g_dLt, Tc, xlo) : 0; z_ref = g_zt[c_lo < Tc ? c_lo : Tc - 1]; }
  const double dB = (double)B;
  double v[4] = { 0., 0., 0., 0. };                        // sw, sw2, sd1, sd2 of this lane
  // STORE: the rare dense redo needs z and w in the workspace (kde_marg_general reads them): same pass, no statistics, no histogram
  auto passes = [&](auto fits_tag, auto store_tag, auto mgA, auto cdfA) {
    constexpr bool FITS = decltype(fits_tag)::value, STORE = decltype(store_tag)::value;
    double* const Hrow_b = Hb + (size_t)(w > 0 ? w - 1 : 0) * HS;
#pragma unroll 1
    for (int tl = tb0; tl < tb1; tl++) {
      const int s = tl * SF_TILE + 2 * lane;
      const double2* tp = tbase + (size_t)tl * (6 * SF_TILE / 2);
      auto ld = [&](const double2* q) { if (NT) { double2 r; r.x = __builtin_nontemporal_load(&q->x); r.y = __builtin_nontemporal_load(&q->y); return r; } return *q; };
      const double2 a = ld(tp), bb = ld(tp + SF_TILE / 2), cc = ld(tp + 2 * SF_TILE / 2), dd = ld(tp + 3 * SF_TILE / 2), ee = ld(tp + 4 * SF_TILE / 2), ff = ld(tp + 5 * SF_TILE / 2);
      const unsigned pq = *reinterpret_cast<const unsigned short*>(pid + s);
      const int pix[2] = { (int)(pq & 255u), (int)(pq >> 8) };
      const double dl[2] = { a.x, a.y }, md1[2] = { bb.x, bb.y }, md2[2] = { cc.x, cc.y }, ipr[2] = { dd.x, dd.y }, l1[2] = { ee.x, ee.y }, l2[2] = { ff.x, ff.y };
      double zz[2], z0[2] = { 0., 0. }, lz0[2] = { 0., 0. };
      bool bad = true;
      if (FITS) {
        z_from_lut_x2(dl[0], dl[1], rec, luts, key0e, nk_e, rlo, ns_e, lmax, Tc, x_last, z_last, zz[0], zz[1], z0[0], z0[1], lz0[0], lz0[1], bad);
        if (wave_any(bad)) { if (bad) { zz[0] = jnp_interp(dl[0], g_dLt, g_zt, Tc, false, 0., 0.); zz[1] = jnp_interp(dl[1], g_dLt, g_zt, Tc, false, 0., 0.); } }
      } else { zz[0] = L.ws_z[so + (s < S ? s : S - 1)]; zz[1] = L.ws_z[so + (s + 1 < S ? s + 1 : S - 1)]; }     // the pre-pass's values
      double wv[2];
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const double z = zz[h];
        const double zp1 = 1. + z;
        const double r = chm_rcp(zp1);
        const double m1 = md1[h] * r, m2 = md2[h] * r;     // m_src = m_det/(1+z) (pop_wrapper.py:70)
        double lz;
        if (FITS) {
          double vv;
          lz = log1pz_from_node(z, z0[h], lz0[h], r, vv);
          const bool nolog = bad || !(vv <= 0.02);
          if (wave_any(nolog)) { if (nolog) lz = chm_log_pos(zp1); }
        } else lz = chm_log_pos(zp1);
        const double wgt = p_m1m2_fused<MASS>(P, m1, m2, l1[h] - lz, l2[h] - lz, mgA, cdfA, ex) * ipr[h];      // w = p_m1m2 / pe_prior (pop_wrapper.py:79)
        wv[h] = wgt;
        if (!STORE && s + h < S) {
          const double d = z - z_ref;
          v[0] += wgt; v[1] += wgt * wgt; v[2] += d; v[3] += d * d;
          const int q = pix[h];
          if (q < Pn) {                                     // binning1d of the masked samples (math.py:32-46, likelihood.py:179-183)
            const double2 pr = *reinterpret_cast<const double2*>(pxp + 2 * q);
            const int idx = bin_index_r(z, lo, pr.x, pr.y, dB);
            double* row = q == pfirst ? Hrow_b : H + (size_t)q * HS;
            atomicAdd(&row[idx], wgt);
          }
        }
      }
      if (STORE) {
        if (s < S) { L.ws_z[so + s] = zz[0]; L.ws_w[so + s] = wv[0]; }
        if (s + 1 < S) { L.ws_z[so + s + 1] = zz[1]; L.ws_w[so + s + 1] = wv[1]; }
      }
    }
  };
  auto run_passes = [&](auto store_tag) {
    if (fast) { if (mass_lds) passes(std::true_type{}, store_tag, mgs, cdfs); else passes(std::true_type{}, store_tag, gm, gc); }
    else { if (mass_lds) passes(std::false_type{}, store_tag, mgs, cdfs); else passes(std::false_type{}, store_tag, gm, gc); }
  };
  run_passes(std::false_type{});
  PH(2);
#pragma unroll
  for (int i = 0; i < 4; i++) v[i] = wave_sum_dpp(v[i]);
  if (lane == 0) { red[4 * w] = v[0]; red[4 * w + 1] = v[1]; red[4 * w + 2] = v[2]; red[4 * w + 3] = v[3]; }
  __syncthreads();
  // ---- boundary rows into their pixels' histograms, in wave order; event statistics by thread 0 meanwhile
  for (int ww = 1; ww < NW; ww++) {
    const int q = bnd[ww];
    if (q >= 0) for (int j = t; j < B; j += NT_) H[(size_t)q * HS + j] += Hb[(size_t)(ww - 1) * HS + j];
  }
  if (t == 0) {
    double sw = 0., sw2 = 0., sd1 = 0., sd2 = 0.;
    for (int i = 0; i < NW; i++) { sw += red[4 * i]; sw2 += red[4 * i + 1]; sd1 += red[4 * i + 2]; sd2 += red[4 * i + 3]; }
    EvStats st;
    const double md = sd1 / (double)S, var = sd2 / (double)S - md * md;      // shifted one-pass std (combine_stats)
    st.zmin = lo; st.zmax = zmx;
    if (sd1 != sd1) { st.zmin = sd1; st.zmax = sd1; }        // jnp.min / jnp.max propagate NaN (any NaN z makes sum(d) NaN)
    st.sd = sqrt(var > 0. ? var : (var != var ? var : 0.));
    st.norm = sw / (double)S; st.n_eff = (sw * sw) / sw2; st.sumw = sw;
    double es[NEVSTAT];
    event_stats_from(L, e, st, es);
#pragma unroll
    for (int i = 0; i < NEVSTAT; i++) esv[i] = es[i];
  }
  __syncthreads();
  const bool ok = esv[4] >= L.pe_neff;                      // likelihood.py:199 (same for every pixel of the event)
  PH(3);
  const bool poisoned = grid_is_poisoned(P.z_bad, zg, Z);
  const int npx = L.neff_pixels[e];
  if (ok) {
    // ---- per-z factors on [k_lo, k_hi], the support of the event's KDE (likelihood.py:270-272, pop_wrapper.py:82-90)
    const int k_lo = ((int)esv[8]) & ~1, k_hi = (int)esv[9];
    const size_t zo = be * Z;
    for (int k = k_lo + t; k <= k_hi; k += NT_) zfactor_point(L, P, e, k, zo, zg, g_zt, g_It, 1, ex);
  }
  __syncthreads();                                          // bkgA / Aw stored; the overlay region is free for the prefix arrays
  PH(4);
  // ---- pixel pass: every wave takes pairs of pixels, one after the other
  {
    const int sub = lane >> 5, sl = lane & 31;
    const int PG = (Pn + 1) / 2;
    double* const Q12 = ov + ((size_t)w * 2 + sub) * 2 * HS;
    double dz[1] = { 0. }, dw[1] = { 0. };
    bool first = true;
    for (int pg = w; pg < PG; pg += NW) {
      const int p = 2 * pg + sub;
      const bool live = p < Pn && p < npx;
      if (!ok || !wave_any(live)) {                            // uniform: every pixel of the event (of this pair: padded pixels) is 0 (or 0 * NaN)
        if (p < Pn && sl == 0) { lik[p] = (live && poisoned) ? __builtin_nan("") : 0.; err[p] = 0.; }
        continue;
      }
      if (!first) wave_sync();
      first = false;
      const int pr = p < Hrows ? p : Hrows - 1;             // (p < Hrows always: Hrows is even)
      const int pp = p < Pn ? p : Pn - 1;
      kde_sub_item<32, 1, BINS, false, NT, true>(L, params, H + (size_t)pr * HS, esv, b, e, p, pp, live, poisoned, 0, 0, dz, dw,
                                                  pxh[pp], Q12, lik + pr, err + pr);
    }
  }
  PH(5);
  __syncthreads();
  // ---- L_i = sum over the pixels in pixel order (likelihood.py:280), rounding bound, dense redo where it matters (k_marg_fixup's rule)
  bool redo = false;
  double x = 0., er = 0.;
  if (w == 0) {
    x = lane < Pn ? lik[lane] : 0.; er = lane < Pn ? err[lane] : 0.;
    const double li = wave_sum(x), es_ = wave_sum(er);
    redo = !L.no_dense && (es_ > D.tol * fabs(li));       // (false for NaN: a NaN event stays NaN)
    const double Li = wave_pixel_sum_regs(x, Pn);
    if (lane == 0) { esv[14] = redo ? 1. : 0.; if (!redo) { L.ev_li[be] = Li; L.ev_ll[be] = log_like_of(Li); } }
  }
  __syncthreads();
  PH(6);
  if (esv[14] == 0.) return;
  // ---- rare: z and w of the event into the workspace, then the pixels above their equal share of the tolerance by the general kernel's
  //      body (dense sums where the bins are light), in pixel order
  stage_tables_ev();
  if (t < NEVSTAT) L.evstat[be * NEVSTAT + t] = esv[t];
  __syncthreads();
  run_passes(std::true_type{});
  __threadfence_block();
  __syncthreads();
  if (w == 0) {
    if (D.redo_count && lane == 0) atomicAdd(D.redo_count, 1);
    const double li = wave_sum(x);
    const double share = D.tol * fabs(li) / (double)Pn;
    double* lp = L.like_pix + be * Pn;
    if (lane < Pn) lp[lane] = x;
    __threadfence_block();
    unsigned long long need = __ballot(lane < Pn && er > share);
    while (need) {
      const int q = __ffsll((long long)need) - 1;
      need &= need - 1ull;
      wave_sync();
      kde_marg_general<true>(L, params, b, e, q, H, true);
    }
    __threadfence_block();
    wave_sync();
    const double y = lane < Pn ? __builtin_nontemporal_load(lp + lane) : 0.;
    const double Li = wave_pixel_sum_regs(y, Pn);
    if (lane == 0) { L.ev_li[be] = Li; L.ev_ll[be] = log_like_of(Li); }
  }
}
