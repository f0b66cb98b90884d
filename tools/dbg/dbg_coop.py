import os, sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from soilmachine_amd.machine import SoilMachine
from soilmachine_amd import capi
if os.environ.get("SMX_LIB"): capi.LIB_PATH = os.environ["SMX_LIB"]
from common import load_cfg
from soilmachine_amd.snapshot import compare
def run(soil, size, ticks, nwater, nwind, dowater, dowind, engine):
    cfg = load_cfg(soil)
    sm = SoilMachine(cfg, dimx=size, dimy=size, seed=0, nwater=nwater, nwind=nwind, dowater=dowater, dowind=dowind, engine=engine)
    sm.tick(ticks, sync=True)
    return sm.map.snapshot(), sm.map.counters()
for (nw, nd, dwa, dw) in [(250, 250, True, False)]:
    for t in (3,):
        a = run('default.soil', 64, t, nw, nd, dwa, dw, capi.ENGINE_SERIAL)
        b = run('default.soil', 64, t, nw, nd, dwa, dw, capi.ENGINE_SPECULATIVE)
        bad = compare(a[0], b[0])
        print(os.environ.get('SMX_SPEC_MODE'), os.environ.get('SMX_SPEC_MODE_WIND'), nw, nd, t, bad or 'equal', {k: b[1][k] for k in b[1] if 'spec' in k or 'nested' in k or 'rand' in k}, 'serial:', {k: a[1][k] for k in a[1] if 'nested' in k or 'rand' in k}, flush=True)
