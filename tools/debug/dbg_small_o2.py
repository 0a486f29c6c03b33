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
params = {k: torch.from_numpy(v).clone() for k, v in c['params'].items()}
ogd = O.build_graph_dict(c['train'], cfg['num_rels'])
ge = {t: torch.from_numpy(v) for t, v in c['global_emb'].items()}
order = sys.argv[1] if len(sys.argv) > 1 else 'so'
for tag in order:
    subject = tag == 's'
    taps = {}
    ops.debug_tap = lambda n, t: taps.setdefault(n, []).append(t.detach().clone())
    with torch.no_grad():
        loss = net(batch, c['hists']['s'], c['hists']['o'], gd, subject=subject)
    ops.debug_tap = None
    lo, parts = O.renet_forward_loss(params, c['batch'], c['hists'][tag][0], c['hists'][tag][1], ogd, ge, cfg['num_rels'],
                                     c['seq_len'], subject=subject, return_parts=True)
    print(tag, 'loss hip', loss.item(), 'oracle', lo.item(), 'gold', float(c['gold']['loss_' + tag]))
    g = net.aggregator.last_batch
    perm_h = g.host.perm
    perm_o = parts['bg'].perm
    print('  perm equal', np.array_equal(perm_h, perm_o))
    for key, okey in (('h_n', 's_h'), ('q_n', 's_q')):
        a = taps[key][0].cpu().numpy(); b = parts[okey].detach().numpy()
        d = np.abs(a - b).max(axis=1)
        print('  ', key, 'max err', d.max(), 'rows>1e-4', np.nonzero(d > 1e-4)[0][:10])
    a = taps['logits'][0].cpu().numpy(); b = parts['ob_pred'].detach().numpy()
    d = np.abs(a - b).max(axis=1)
    print('   logits max err', d.max(), 'rows', np.nonzero(d > 1e-3)[0][:10])
    a = taps['logits'][1].cpu().numpy(); b = parts['ob_pred_r'].detach().numpy()
    d = np.abs(a - b).max(axis=1)
    print('   logits_r max err', d.max(), 'rows', np.nonzero(d > 1e-3)[0][:10])
    print('   loss parts oracle', parts['loss_sub'].item(), parts['loss_r'].item())
