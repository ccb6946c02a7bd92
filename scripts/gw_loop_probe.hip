// gw_loop_probe.hip -- what do the bodies of the two hot kernels sustain on the card with HBM out of the picture?  (diagnostic, not product)
//
// Compiled INTO the library by -DCHM_PROBE builds only (chimera_hip.hip includes this file and scripts/sample_body_probe.hip at its end:
// scripts/build_variant.sh probe -DCHM_PROBE); driven by scripts/run_probes.py -> profiles/r06/probe_ceilings.json, which bench.py reads for
// roofline.frac_of_sustained.  Round 5 priced the kernels against MIN_INST, a count made on paper; this replaces it with a measurement.
//
// k_probe_gw: every one-wave block runs kde_marg_sub2_body<32, 4, 200, false> -- the production body of k_kde_marg_sub2, not a copy -- once
// on one of the (draw, pixel group, event) items of a SMALL resident workload (a handful of events x draws: (z, w), p_cat rows, per-z factors
// and event statistics of a preceding chm_eval, a few MB in all), so that after the first touch every load is served by L2 / the memory-side
// cache: set-up, histogram, prefix sums, bandwidth, grid loop and final scans at the kernel's own occupancy (4 waves per SIMD, 9.6 KB of LDS per
// wave), with their LDS traffic and bank conflicts, without the HBM round trips.  The launches are repeated for >= 1 s so that the board settles at
// the clock it holds under this instruction stream.  Results stored by different waves coincide (same item, same value).
// No loop around the body: a launch is nblocks x reps one-wave blocks, each of which makes ONE body call, as a production block does.  (The first
// form of this probe looped the body `reps` times inside a block; holding the kernel arguments and the loop state across the body cost the register
// allocator 12 vector + 29 scalar spilled registers here and 48 + 82 in the sample probe -- scratch traffic the production kernels do not have.  A
// kernel that calls the body once compiles to the production kernel's own code, and the launch of the many short blocks is part of what is measured.)
template <int IPW>
__global__ void __launch_bounds__(64, 4) k_probe_gw(LikeDev L, const DevParams* params, int ny) {
  extern __shared__ double lds_all[];
  const unsigned q = blockIdx.x * 7919u + blockIdx.y * 104729u;       // neighbouring waves work on unrelated items, as in a production launch
  const int bx = (int)(q % (unsigned)L.nb), by = (int)((q / (unsigned)L.nb) % (unsigned)ny), bz = (int)((q / (unsigned)L.nb / (unsigned)ny) % (unsigned)L.E_cnt);
  CLK_BEGIN;
  kde_marg_sub2_body<32, IPW, 200, false>(L, params, bx, by, bz, lds_all);
  CLK_END(0);
}

// which: 0 = GW kernel body (IPW 4), 1 = sample-stage body; `launches` launches of `nblocks` x `reps` blocks, one body call each; ms[i] = HIP-event time of launch i
extern "C" int chm_debug_probe(chm_like* like, int32_t which, int32_t nb, int32_t nblocks, int32_t reps, int32_t launches, double* ms) {
  if (!like || !ms || nb < 1 || nb > like->nb_ws || nblocks < 1 || reps < 1 || reps > 65535 || launches < 1) return fail(CHM_E_ARG, "chm_debug_probe: bad argument (nb must not exceed the draws of the preceding chm_eval)");
  Ctx& c = like->ctx;
  HIPCHK(hipSetDevice(c.device));
  HIPCHK(hipStreamSynchronize(c.stream));
  LikeDev L = like->L;
  L.e_off = 0; L.E_cnt = L.E; L.nb = nb; L.p_gw_dump = nullptr; L.no_dense = 0; L.ev_publish = 0;
  L.zw_stream = 0;                                          // (plain stores: with the streaming hint of a C3-sized launch the probe's few MB would be written out to HBM over and over)
  L.zg_i = like->d_zg_i; L.zg_t = like->d_zg_t; L.zg_lz = like->d_zg_lz;
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  for (int i = 0; i < launches; i++) {
    HIPCHK(hipEventRecord(e0, c.stream));
    if (which == 0) {
      if (L.mode != CHM_MODE_MARG || L.num_bins != 200 || !L.binning || !L.has_cut || (L.Z & 1)) return fail(CHM_E_ARG, "chm_debug_probe: the GW probe needs the standard marginalized configuration");
      const int PG2 = (L.P + 1) / 2, ny = (PG2 + 3) / 4;
      const size_t lds_sub = sizeof(double) * (5 * (200 + 1 + 7 + 1) + 201);      // (the wave's slice of the production launch: chm_eval)
      hipLaunchKernelGGL((k_probe_gw<4>), dim3(nblocks, reps), dim3(64), lds_sub, c.stream, L, (const DevParams*)c.d_params, ny);
    } else {
      if (!like->probe_lds_fast) return fail(CHM_E_ARG, "chm_debug_probe: the preceding chm_eval did not take the fast sample stage");
      SampFast F = like->F; F.lut = like->probe_lut;
      allow_lds((k_probe_samples<2>), like->probe_lds_fast);
      hipLaunchKernelGGL((k_probe_samples<2>), dim3(nblocks, reps), dim3(64 * CHM_SF_WAVES), like->probe_lds_fast, c.stream, L, F, (const DevParams*)c.d_params,
                         (const double*)c.zt, (const double*)c.dLt, (const double*)c.mg, (const double*)c.cdf, (const double*)c.rec, c.TcMax, c.TmMax);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e1, c.stream));
    HIPCHK(hipStreamSynchronize(c.stream));
    float t = 0.f;
    HIPCHK(hipEventElapsedTime(&t, e0, e1));
    ms[i] = t;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return CHM_OK;
}

// The shader clock DURING other work: one wave reads the core-clock counter (s_memtime) and the constant 100 MHz counter (s_memrealtime) `usec`
// microseconds apart, on a stream of its own, while the caller's evaluations run on theirs -- ghz[0] = core cycles per nanosecond over that window.
__global__ void __launch_bounds__(64) k_probe_clock(long long ticks, unsigned long long* out) {
  const unsigned long long r0 = wall_clock64(), c0 = clock64();
  unsigned long long r1 = r0;
  while ((long long)(r1 - r0) < ticks) { __builtin_amdgcn_s_sleep(32); r1 = wall_clock64(); }
  const unsigned long long c1 = clock64();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
}
extern "C" int chm_debug_clock(chm_like* like, int32_t usec, double* ghz) {
  if (!like || !ghz || usec < 1 || usec > 1000000) return fail(CHM_E_ARG, "chm_debug_clock: bad argument");
  HIPCHK(hipSetDevice(like->ctx.device));
  static thread_local hipStream_t s = nullptr;
  static thread_local unsigned long long* h = nullptr;
  if (!s) { HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); HIPCHK(hipHostMalloc((void**)&h, 16, hipHostMallocDefault)); }
  hipLaunchKernelGGL(k_probe_clock, dim3(1), dim3(64), 0, s, (long long)usec * 100, h);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(s));
  ghz[0] = h[1] ? (double)h[0] / ((double)h[1] * 10.) : 0.;       // 100 MHz ticks: 10 ns each
  return CHM_OK;
}

// g_clk (chm_kernels.h): out = { GW kernel cycles, its 10 ns ticks, sample stage cycles, ticks } summed since the last call; cleared
#ifdef CHM_CLOCK_STAMP
extern "C" int chm_debug_clock_stamps(double out[4]) {
  unsigned long long h[4], z[4] = { 0, 0, 0, 0 };
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_clk), sizeof(h)));
  for (int i = 0; i < 4; i++) out[i] = (double)h[i];
  HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_clk), z, sizeof(z)));
  return CHM_OK;
}
#endif
