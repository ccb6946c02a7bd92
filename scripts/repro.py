import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import helpers as H
from tests.golden.make_golden import CASES, LAMBDAS
names = sys.argv[1:] or sorted(CASES)
keep = []
for name in names:
    pixelated, kind, models, like_kw = CASES[name]
    ev, inj, exp = H.load_golden(name)
    like, pop, _ = H.build_product(ev, inj, pixelated=pixelated, kind=kind or 'marginalized', models=models, like_kw=like_kw)
    for i, lam in enumerate(LAMBDAS):
        print(name, i, 'eval...', flush=True)
        r = like.compute_all(**lam)
        print('   ', r[3], exp['log_hyper'][i], flush=True)
    p0 = like.population.update(**LAMBDAS[0])
    gp = like.p_gw3d(p0) if pixelated else like.p_gw1d(p0)
    print(name, 'pgw ok', flush=True)
    if os.environ.get('KEEP'): keep.append(like)
