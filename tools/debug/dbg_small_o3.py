import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 're-net_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import train_case, O
import graph as G, renet_hip as K, model as M, utils as U, ops
dev = torch.device('cuda:0')
c = train_case('small', 200)
cfg = c['cfg']
net = M.RENet(cfg['num_ent'], 200, cfg['num_rels'], dropout=0.0, seq_len=c['seq_len'])
net.load_state_dict({k: torch.from_numpy(v) for k, v in c['params'].items()})
net.global_emb = {t: torch.from_numpy(v).view(1, 1, -1) for t, v in c['global_emb'].items()}
net.to(dev).eval()
gd = U.build_graph_dict(c['train'], cfg['num_rels'])
batch = torch.from_numpy(c['batch']).to(dev)
for rep in range(2):
    for tag in 'so':
        taps = {}
        ops.debug_tap = lambda n, t: taps.setdefault(n, []).append(t.detach().clone())
        loss = net(batch, c['hists']['s'], c['hists']['o'], gd, subject=(tag == 's'))
        ops.debug_tap = None
        lg = taps['logits'][0]
        tgt = torch.from_numpy(c['batch'][:, 2 if tag == 's' else 0][net.aggregator.last_batch.host.perm]).to(dev)
        ref_ce = torch.nn.functional.cross_entropy(lg, tgt.long())
        lgr = taps['logits'][1]
        tr = torch.from_numpy(c['batch'][:, 1][net.aggregator.last_batch.host.perm]).to(dev)
        ref_r = torch.nn.functional.cross_entropy(lgr, tr.long())
        print(rep, tag, 'loss', loss.item(), 'gold', float(c['gold']['loss_' + tag]), 'torch CE on tapped logits', (ref_ce + 0.1 * ref_r).item(),
              ref_ce.item(), ref_r.item(), 'B', lg.shape, lgr.shape)
