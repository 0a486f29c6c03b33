#!/usr/bin/env python
"""cProfile of ONE inference advance (model.py:222-328) at num_k 1000 on the ICEWS18-shaped stream (GPU only): where the
host time of RENet._advance_time goes.   python tools/advance_profile.py [top_n]"""
import cProfile
import os
import pstats
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
sys.path.insert(0, os.path.join(ROOT, 're-net_amd'))
import infer_bench


def main():
    top = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    dev = torch.device('cuda:0')
    net, gnet, te, tes, teo, total = infer_bench.setup('ICEWS18', 3, 200, dev, num_k=1000)
    ts = np.unique(te[:, 3])
    with torch.no_grad():
        net._advance_time(torch.tensor(int(ts[1])), gnet)          # warm-up (allocator, first-use costs)
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        pr.enable()
        net._advance_time(torch.tensor(int(ts[2])), gnet)
        torch.cuda.synchronize()
        pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats('cumulative').print_stats(top)
    st.sort_stats('tottime').print_stats(top)


if __name__ == '__main__':
    main()
