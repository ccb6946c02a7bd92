#!/usr/bin/env python3
"""The shader clock the card holds DURING the production step and during the body probes (-DCHM_PROBE build; chm_debug_clock: one wave on a stream of
its own compares the core-clock counter with the constant 100 MHz counter over 200 us windows while the work runs).

  scripts/build_variant.sh clock -DCHM_PROBE -DCHM_CLOCK_STAMP; CHIMERA_LIB=chimera_amd/lib/variants/libchimera_hip_clock.so python3 scripts/clock_under_load.py [--events 1000] [--seconds 4]
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--events', type=int, default=1000)
  ap.add_argument('--inj', type=int, default=100000)
  ap.add_argument('--draws', type=int, default=128)
  ap.add_argument('--seconds', type=float, default=4.)
  args = ap.parse_args()
  import chimera_amd as CH
  from chimera_amd import synth, _lib
  from chimera_amd.catalog import dVdz_completeness, pixelated_catalog
  L = _lib.lib()
  if not hasattr(L, 'chm_debug_clock') or not hasattr(L, 'chm_debug_clock_stamps'):
    raise SystemExit('clock_under_load.py: the loaded library has no chm_debug_clock (scripts/build_variant.sh clock -DCHM_PROBE -DCHM_CLOCK_STAMP, CHIMERA_LIB=...)')
  L.chm_debug_clock.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_double)]
  L.chm_debug_clock.restype = C.c_int
  L.chm_debug_probe.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double)]
  L.chm_debug_probe.restype = C.c_int
  L.chm_debug_clock_stamps.argtypes = [C.POINTER(C.c_double)]
  L.chm_debug_clock_stamps.restype = C.c_int

  def stamps():
    o = (C.c_double * 4)()
    _lib.check(L.chm_debug_clock_stamps(o))
    return {"gw_kernel_GHz": o[0] / (o[1] * 10.) if o[1] else None, "sample_stage_GHz": o[2] / (o[3] * 10.) if o[3] else None,
            "gw_kernel_wave_us": o[1] * 1e-2, "sample_stage_wave_us": o[3] * 1e-2}
  cfg, ev, inj = synth.make_config('C3', E=args.events, I=args.inj)
  pe_fields = ('m1det', 'm2det', 'dL', 'ra', 'dec', 'pe_prior', 'pixels_opt_nsides', 'ra_pix', 'dec_pix', 'gw_loc2d_pdf', 'pixels_pe_opt_nside')
  th = CH.data.theta_pe_det(**{k: ev[k] for k in pe_fields})
  gal_cat = pixelated_catalog(dVdz_completeness(z_range=[0.073, 1.3]), p_cat=ev['p_cat'], z_grids=ev['z_grids'], neff_pixels=ev['neff_pixels'])
  pop = CH.population(CH.cosmo.flrw(H0=70., Om0=0.25, z_max=5.), CH.mass.plp(), CH.rate.madau_dickinson(gamma=2.7, kappa=3., zp=2.), gal_cat=gal_cat, scale_free=True)
  sel = CH.selection_function(CH.data.theta_inj_det(**{k: inj[k] for k in ('m1det', 'm2det', 'dL', 'p_draw')}), N_inj=inj['N_inj'], N_eff=5.)
  like = CH.hyperlikelihood(th, ev['z_grids'], pop, sel, kind_p_gw3d='marginalized', kernel='epan', bw_method=None, cut_grid=2, binning=True, num_bins=200)
  h = like._handle()
  draws = dict(H0=np.linspace(62., 78., args.draws))
  like.batch(draws)

  def sample(stop, out):
    g = C.c_double()
    while not stop.is_set():
      _lib.check(L.chm_debug_clock(h, 200, C.byref(g)))
      out.append(g.value)
      time.sleep(0.002)

  def watched(work):
    stop, out = threading.Event(), []
    t = threading.Thread(target=sample, args=(stop, out)); t.start()
    t0 = time.perf_counter(); n = work(); dt = time.perf_counter() - t0
    stop.set(); t.join()
    a = np.array(out[len(out) // 4:])                       # the first quarter: ramp
    return {"samples": len(out), "GHz_median": float(np.median(a)), "GHz_q10": float(np.quantile(a, .1)), "GHz_q90": float(np.quantile(a, .9)), "seconds": dt, "units": n}

  res = {"workload": {"E": args.events, "draws": args.draws}}

  def idle():
    time.sleep(1.); return 0
  res["idle"] = watched(idle)

  def steps():
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < args.seconds:
      like.batch(draws); n += 1
    return n
  stamps()
  r = watched(steps); r["ms_per_step"] = r["seconds"] / r["units"] * 1e3
  r["inside_the_kernels"] = stamps()
  res["production_steps"] = r
  nbp = min(args.draws, 4)
  # (the probes want a cache-resident workload: --events 4 --draws 16)
  for which, name, nblocks in ((0, 'probe_gw', 256 * 16 * 4), (1, 'probe_samples', 256 * 4 * 2)) if args.events <= 16 else ():
    ms = (C.c_double * 64)()
    reps = 40 if which == 0 else 4
    _lib.check(L.chm_debug_probe(h, which, nbp, nblocks, reps, 1, ms))
    reps = min(65535, max(1, int(reps * 150. / ms[0])))
    nl = min(64, max(4, int(np.ceil(args.seconds / 0.15))))

    def probe():
      _lib.check(L.chm_debug_probe(h, which, nbp, nblocks, reps, nl, ms)); return nl
    stamps()
    res[name] = watched(probe)
    res[name]["inside_the_kernels"] = stamps()
  print(json.dumps(res, indent=1))
  like.close(); sel.close()


if __name__ == '__main__':
  main()
