import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from chimera_amd import synth
from tests import helpers as H
cfg, ev, inj = synth.make_config('C3', seed=20250926, E=100, I=100000)
like, _, _ = H.build_product(ev, inj)
like.set_option('graph_max_nb', 0)
for h in (66., 67., 68., 69., 70.):
  like(H0=h)
