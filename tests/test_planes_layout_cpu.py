"""CPU model of the planes GEMM's data path (tools/p6_layout_sim.py): the T16 plane format -> LDS-DMA pieces -> LDS image
-> MFMA fragments, for both consumer roles, and the LDS bank-conflict count of the fragment reads.  No GPU needed: the
model is the specification the kernel (csrc/gemm_p6.h) and the producers (renet_pack_planes, renet_softmax_ce_planes)
implement; the GPU tests (tests/test_gpu_planes.py) check the kernels against fp64 products."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))


def test_t16_fragments_are_correct_and_conflict_free_in_both_roles():
    import p6_layout_sim as sim
    b128_extra, tr_extra = sim.check()
    assert b128_extra == 0 and tr_extra == 0


def test_t16_offset_matches_the_python_binding_helper():
    import numpy as np
    import torch
    import p6_layout_sim as sim
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 're-net_amd'))
    import renet_hip as K
    rp, cp = 256, 512
    dense = torch.arange(rp * cp, dtype=torch.float32).reshape(rp, cp)
    tiled = torch.from_numpy(sim.to_t16(dense.numpy().astype(np.int64)).astype(np.float32))
    m = K.PlanesMat(tiled.to(torch.bfloat16).reshape(1, rp, cp).repeat(3, 1, 1), rp, cp)
    # bf16 cannot hold the indices exactly: compare through the same rounding
    back = K.planes_to_dense(m)[0]
    assert torch.equal(back, dense.to(torch.bfloat16).float())
