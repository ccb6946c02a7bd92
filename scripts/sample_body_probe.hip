// sample_body_probe.hip -- the body of the fast sample stage on a cache-resident workload (see scripts/gw_loop_probe.hip; -DCHM_PROBE builds only)
//
// k_probe_samples: every block of 256 threads runs samples_fast_body<MASS, false, false> -- the production body of k_samples_fast: table staging into
// LDS, then the chunks of 4096 samples of ALL the events of a small resident workload (nbx = 1: with four events a body call is what a production
// block does, four chunks behind one staging) -- once, for one of the draws (a launch is nblocks x reps such blocks; no loop around
// the body, see scripts/gw_loop_probe.hip).  Tiles are read from L2, (z, w) and the partial records
// are written to the same few MB over and over (write-combined in L2): z(dL) through the direct-index table and the node records, log(1 + z),
// p_m1m2_fused with its table exponentials and LDS gathers, the per-wave statistics -- without waiting for HBM.
template <int MASS>
__global__ void __launch_bounds__(64 * CHM_SF_WAVES, CHM_SF_MINW) k_probe_samples(LikeDev L, SampFast F, const DevParams* params, const double* zt_all,
                                                                                const double* dLt_all, const double* mg_all, const double* cdf_all,
                                                                                const double* rec_all, int TcMax, int TmMax) {
  extern __shared__ double lds[];
  const int b = (int)((blockIdx.x + blockIdx.y) % (unsigned)L.nb);
  CLK_BEGIN;
  samples_fast_body<MASS, false, false>(L, F, params, zt_all, dLt_all, mg_all, cdf_all, rec_all, TcMax, TmMax, b, 0, 1, lds);
  CLK_END(2);
}
