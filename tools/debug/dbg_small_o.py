import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 're-net_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import train_case
import graph as G, renet_hip as K, model as M, utils as U
dev = torch.device('cuda:0')
c = train_case('small', 200)
cfg = c['cfg']
net = M.RENet(cfg['num_ent'], 200, cfg['num_rels'], dropout=0.0, seq_len=c['seq_len'])
net.load_state_dict({k: torch.from_numpy(v) for k, v in c['params'].items()})
net.global_emb = {t: torch.from_numpy(v).view(1, 1, -1) for t, v in c['global_emb'].items()}
net.to(dev).eval()
gd = U.build_graph_dict(c['train'], cfg['num_rels'])
batch = c['batch']
for tag, subject in (('s', True), ('o', False)):
    s, r, o, rel, reverse = net._direction(batch, subject)
    g = net.aggregator.build(c['hists'][tag], s, r, net.ent_embeds, gd, net.global_emb, True)
    hb = g.host
    deg = np.diff(g.row_ptr.cpu().numpy())
    print(tag, 'N', g.N, 'E', g.E, 'nA', g.nA, 'maxdeg', deg.max(), 'hubs', 0 if g.heavy_rows is None else g.heavy_rows.numel(),
          'groups', g.n_groups, g.n_groups_out, 'grp sizes max', int(np.diff(g.grp_ptr.cpu().numpy()).max()))
    h0 = net.ent_embeds[g.node_ent.long()].contiguous()
    T = 2 * cfg['num_rels']
    shift = T // 2 if reverse else 0
    w1, l1 = net.aggregator.rgcn1.weight, net.aggregator.rgcn1.loop_weight
    for pruned in (False, True):
        n_out = g.nA if pruned else g.N
        ad = K.gemm(h0[:n_out], l1)
        a = ad.clone(); b = ad.clone()
        K.rgcn_gather_items(h0, g, w1, shift, False, a, 0.0, 0, True, a, use_norm=True, pruned=pruned)
        hv = g.heavy_rows_out if pruned else g.heavy_rows
        K.rgcn_gather(h0, g.row_ptr, g.col, g.etype, g.norm, w1, shift, False, b, 0.0, 0, True, b, hv, g.heavy_thresh)
        d = (a - b).abs().max(dim=1).values.cpu().numpy()
        bad = np.nonzero(d > 1e-4)[0]
        print('  pruned', pruned, 'max diff', d.max(), 'bad rows', bad[:20], 'deg of bad', deg[bad[:20]])
        if len(bad):
            it_src, it_type, gp = g.it_src.cpu().numpy(), g.it_type.cpu().numpy(), g.grp_ptr.cpu().numpy()
            fl = np.nonzero((it_type == -1) & np.isin(it_src, bad[:3]))[0]
            for f in fl:
                gi = np.searchsorted(gp, f, 'right') - 1
                print('   row', it_src[f], 'flush at item', f, 'group', gi, 'range', gp[gi], gp[gi + 1], 'n', gp[gi + 1] - gp[gi],
                      'pos in group', f - gp[gi])
