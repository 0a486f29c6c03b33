#!/usr/bin/env python
"""Benchmark of the RE-Net hot path on MI355X: one "step" = one full training step on one batch of
1024 quadruples -- both directions (train.py:136-137) of RGCN x2 -> sequence assembly -> GRU x2 ->
score heads -> loss, backward, gradient all-reduce (N>1), clip (train.py:140) and Adam.

Workload (BASELINE.json configs[1]): ICEWS18-shaped synthetic stream (re-net_amd/synth.py, seed 999),
n_hidden=200, seq_len=10, batch=1024 per GPU, dropout 0.5, fp32, random-init weights.
Inputs are device-resident when the timed region starts: the batch graphs / packed layouts of the
W+K steps are prepared (host builder + one upload each) before it; the per-step host build time is
reported separately (`host_build_ms`) and an end-to-end rate with the builder in the loop as
`e2e_value`.  Launch: python bench.py [--gpus N --steps K --warmup W].  For N > 1 either under
torch.distributed.run (one rank per GPU, RCCL: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment) or
plainly -- without WORLD_SIZE in the environment the script re-executes itself under torch.distributed.run with N
ranks on 127.0.0.1.  Rank 0 prints ONE JSON line.

Precision (round 4): `value` is timed in the DEFAULT fp32-class GEMM mode, bf16x6 (fp32 storage, every operand split
into three bf16 planes = all 24 significand bits, six MFMA products, fp32 accumulation).  The two other modes run as
child processes of the same command and are reported as full records `value_f16x3` (22-bit operands, the fast mode)
and `value_exact_f32` (exact fp32 MFMA products), each with its own kernels / gemm_shapes / roofline.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, 're-net_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3       # MI355X_MICROARCH.md: f32-input MFMA dense peak
MFMA_BF16_PEAK_TF = 2500.0     # MI355X_MICROARCH.md: bf16 MFMA dense peak (never the 2:1-sparse figure)


def _r(v, sig=4):
    """floats to `sig` significant digits (the line is for reading and parsing, the detail file keeps full precision)"""
    if isinstance(v, float):
        return float('%.*g' % (sig, v))
    return v


def _pick(d, keys, sig=4):
    return {k: _r(d[k], sig) for k in keys if d is not None and k in d and d[k] is not None} if d else None


LINE_LIMIT = 4096       # the driver keeps ~8 KB of stdout tail; round 4's 37 KB line was cut and recorded as parsed: null


def compact_line(out, detail_path=None):
    """The ONE stdout line of a run, from the full record `out` (which goes to bench_detail.json): the contract's keys,
    the rooflines, the CPU baseline, parity, and one-line summaries of the companion modes / other configs / scaling
    modes.  Guaranteed < LINE_LIMIT bytes: optional blocks are dropped from the end of `optional` until it fits
    (tests/test_host_cpu.py runs this on recorded runs, N = 1 and N > 1)."""
    roof = out.get('roofline')
    line = {
        'metric': out['metric'], 'value': _r(out['value'], 6), 'unit': out['unit'], 'n_gpus': out['n_gpus'],
        'steps': out['steps'], 'warmup': out['warmup'], 'ms_per_step': _r(out['ms_per_step'], 5),
        'higher_is_better': True, 'scaling': out['scaling'], 'vs_baseline': out.get('vs_baseline'),
        'dtype': out['dtype'], 'data': out['data'], 'gemm_mode': out.get('gemm_mode'),
        'config': {'workload': out['config']['workload'], 'parallelism': out['config'].get('parallelism'),
                   'global_batch': out.get('global_batch'),
                   'nodes': (out['config'].get('batch_graph') or {}).get('nodes'),
                   'edges': (out['config'].get('batch_graph') or {}).get('edges')},
        'roofline': _pick(roof, ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic',
                                 'operand_bytes_per_launch', 'avg_us', 'calls_per_step')),
        'cpu_baseline': _pick(out.get('cpu_baseline'), ('value', 'unit', 'cores', 'kind', 'ms_per_step', 'sample')),
        'parity': _pick(out.get('parity'), ('rel_err', 'tolerance', 'grad_rel_err', 'grad_tolerance', 'batches')),
    }
    if roof is not None and 'traffic' not in line['roofline']:
        line['roofline']['traffic'] = None
    optional = []
    g = out.get('roofline_rgcn_gather') or {}
    if g:
        optional.append(('roofline_rgcn_gather',
                         {k.replace('rgcn_gather_', ''): _pick(v, ('frac_strict', 'frac', 'avg_us', 'traffic'), 3)
                          for k, v in g.items()}))
    if out.get('roofline_gru'):
        optional.append(('roofline_gru', _pick(out['roofline_gru'], ('bound', 'achieved', 'peak', 'unit', 'frac', 'avg_us',
                                                                     'calls_per_step'))))
    if out.get('kernel_only'):
        optional.append(('kernel_only', _pick(out['kernel_only'], ('ms_per_step', 'glue_ms_per_step',
                                                                   'timed_pass_ms_per_step'))))
    if out.get('value_median'):
        optional.insert(0, ('value_median', _pick(out['value_median'], ('value', 'ms_per_step', 'steps', 'p10_ms', 'p90_ms'))))
    if out.get('value_list_api'):
        optional.append(('value_list_api', _pick(out['value_list_api'], ('value', 'ms_per_step', 'ms_per_step_median', 'steps'))))
    for k in ('e2e_device_builder', 'host_enqueue_ms_per_step', 'plan_entries_per_step', 'launches_per_step', 'host_build_ms',
              'last_loss', 'rccl_ranks_seen', 'scaling_mode'):
        if out.get(k) is not None:
            optional.append((k, _r(out[k])))
    if out.get('encoder_only'):
        optional.append(('encoder_only', _pick(out['encoder_only'], ('value', 'ms_per_step'))))
    for sc in ('scaling_weak', 'scaling_strong', 'scaling_exact'):
        if out.get(sc):
            optional.append((sc, _pick(out[sc], ('value', 'ms_per_step', 'global_batch', 'batch_per_gpu', 'last_loss'))))

    def summary(rec):
        if not rec:
            return None
        if 'error' in rec:
            return {'error': str(rec['error'])[-80:]}
        s_ = _pick(rec, ('value', 'ms_per_step', 'dtype', 'gemm_mode'))
        if rec.get('roofline'):
            s_['roofline_frac'] = _r(rec['roofline'].get('frac'), 3)
        if rec.get('parity'):
            s_['rel_err'] = _r(rec['parity'].get('rel_err'), 2)
            s_['grad_rel_err'] = _r(rec['parity'].get('grad_rel_err'), 2)
        return s_
    modes = {k: summary(out.get(k)) for k in ('value_bf16x6', 'value_f16x3', 'value_exact_f32') if out.get(k)}
    if modes:
        optional.append(('modes', modes))
    if out.get('other_configs'):
        optional.append(('other_configs', {k: summary(v) for k, v in out['other_configs'].items()}))
    if detail_path:
        line['detail'] = detail_path
    for k, v in optional:
        line[k] = v
    text = json.dumps(line, separators=(',', ':'))
    while len(text) >= LINE_LIMIT and optional:
        k, _ = optional.pop()
        line.pop(k, None)
        line['dropped'] = line.get('dropped', 0) + 1
        text = json.dumps(line, separators=(',', ':'))
    # the mandatory part alone over the limit (a very long workload / sample / metric string): shorten its free-text fields
    # rather than lose the line after the whole benchmark has run
    for holder, key in ((line.get('cpu_baseline') or {}, 'sample'), (line['config'], 'workload'), (line, 'metric'),
                        (line, 'detail'), (line, 'data')):
        if len(text) < LINE_LIMIT:
            break
        v = holder.get(key)
        if isinstance(v, str) and len(v) > 48:
            holder[key] = v[:45] + '...'
            line['truncated'] = line.get('truncated', 0) + 1
            text = json.dumps(line, separators=(',', ':'))
    if len(text) >= LINE_LIMIT:
        text = json.dumps({k: line[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step',
                                                'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data')},
                          separators=(',', ':'))[:LINE_LIMIT - 1]
    return text


def write_detail(out):
    """The full record (kernel tables, gemm_shapes, complete companion / other-config records) -> bench_detail.json next
    to this script and, on a gpurun box, under gpurun_out/ (which travels back).  Returns the path named in the line."""
    name = 'bench_detail.json' if out.get('n_gpus', 1) == 1 else 'bench_detail_n%d.json' % out['n_gpus']
    for d_ in (ROOT, os.path.join(ROOT, 'gpurun_out')):
        try:
            if os.path.isdir(d_):
                with open(os.path.join(d_, name), 'w') as f:
                    json.dump(out, f)
        except OSError:
            pass
    return name


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--shape', default='ICEWS18')
    ap.add_argument('--batch', type=int, default=1024)
    ap.add_argument('--hidden', type=int, default=200)
    ap.add_argument('--seq-len', type=int, default=10)
    ap.add_argument('--dropout', type=float, default=0.5)
    ap.add_argument('--cpu-steps', type=int, default=10,
                    help='oracle steps timed for cpu_baseline after --cpu-warmup untimed ones (0 = skip; ~3 s each on '
                         'the GPU box with 32 threads; SURVEY 8d: >= 10 steps after 2 warm-ups, medians)')
    ap.add_argument('--cpu-warmup', type=int, default=2)
    ap.add_argument('--enc-steps', type=int, default=50,
                    help='steps of the encoder-only companion measurement (RGCN x2 + sequence assembly + GRU x2, '
                         'forward + backward, no heads / clip / Adam; 0 = skip)')
    ap.add_argument('--companions', type=int, default=1,
                    help='N > 1 only: also time the strong and exact scaling modes (same K steps each) and report '
                         'them as scaling_strong / scaling_exact in the same JSON line (0 = skip)')
    ap.add_argument('--cpu-threads', type=int, default=0, help='torch threads for the CPU baseline (0 = min(32, cores))')
    ap.add_argument('--e2e-steps', type=int, default=5)
    ap.add_argument('--list-steps', type=int, default=30,
                    help="steps of `value_list_api`: the reference's own loop body (train.py:133-142: nested history lists, two "
                         'model() calls, loss.backward(), clip_grad_norm_, torch.optim.Adam, loss.item()) timed end to end '
                         '(0 = skip)')
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong', 'exact'],
                    help='weak: --batch quadruples per GPU (default); strong: --batch quadruples per step in total, '
                         'batch / N per GPU, each rank with its own (smaller) reference batch; exact: ONE reference '
                         'batch of --batch quadruples per step, every rank builds its graph and keeps 1/N of the '
                         'sequences, gradients are summed (SURVEY 8e option (i): N-GPU step == 1-GPU step)')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'],
                    help="f32: fp32-class GEMMs (f16x3 split; RENET_GEMM=bf16x6 | f32 select the 24-bit split / exact fp32); bf16: GEMM operands "
                         "rounded to bf16, fp32 accumulate (BASELINE config 5)")
    ap.add_argument('--passes', choices=('merged', 'pair', 'serial'), default='merged',
                    help='how a step runs the subject and the object pass of its batch (train.py:136-138): merged = '
                         'one pass over the 2B sequences (RENet.loss_prepared_both), pair = two passes with their four '
                         'GRU recurrences sharing launches (loss_prepared_pair), serial = RENet.loss_prepared twice')
    ap.add_argument('--no-pair', action='store_true',
                    help='run the subject and object passes strictly one after the other (RENet.loss_prepared twice) '
                         'instead of RENet.loss_prepared_pair')
    ap.add_argument('--f32-steps', type=int, default=40,
                    help='steps of the companion runs in the OTHER fp32-class GEMM modes (of bf16x6 / f16x3 / f32; child '
                         'processes, full records; 0 = skip)')
    ap.add_argument('--other-steps', type=int, default=20,
                    help='steps of the other BASELINE.json configs (WIKI-shaped, GDELT-shaped, YAGO-shaped n_hidden 400 '
                         'seq_len 15 bf16 storage), each a child process reported under `other_configs` (0 = skip)')
    ap.add_argument('--plain', action='store_true',
                    help='only the W + K steps of `value` (no event-timed second pass, no companions, no CPU baseline): '
                         'the command to put under rocprofv3 --kernel-trace --stats, so that the trace holds the product '
                         'configuration alone')
    ap.add_argument('--child', default='', help='(internal) this run is a companion of another: print the reduced record')
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it: re-execute under torch.distributed.run, N ranks on this
    node, rendezvous on 127.0.0.1 (the container hostname may not resolve).  Returns the child's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.plain:
        args.cpu_steps = args.enc_steps = args.e2e_steps = args.f32_steps = args.other_steps = args.companions = 0
        args.list_steps = 0
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))          # no launcher around us: start the N ranks ourselves
    if world != args.gpus and rank == 0:
        sys.stderr.write('WORLD_SIZE (%d) != --gpus (%d): running with %d ranks\n' % (world, args.gpus, world))
    if os.environ.get('RENET_BENCH_LAUNCH_CHECK') == '1':
        # launcher self-test (tests/test_parallel_cpu.py, no GPU needed): the ranks started above rendezvous over gloo,
        # all-reduce a one, and rank 0 prints the line's launch fields; nothing is measured
        if world > 1:
            dist.init_process_group('gloo')
            ones = torch.ones(1)
            dist.all_reduce(ones)
            seen = int(ones.item())
            dist.barrier()
            dist.destroy_process_group()
        else:
            seen = 1
        if rank == 0:
            print(json.dumps({'launcher_check': True, 'n_gpus': world, 'rccl_ranks_seen': seen, 'gpus_arg': args.gpus}))
        return
    assert torch.cuda.is_available(), 'bench.py needs a HIP device (no CPU fallback exists)'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    ranks_seen = 1
    force_rccl = os.environ.get('RENET_FORCE_REDUCER') == '1'
    if world > 1 or force_rccl:
        # RENET_FORCE_REDUCER=1 on a one-GPU box: a ONE-rank RCCL group, so that the reducer's call sequence (early bucket
        # on its side stream, tail buckets, waits) really goes through RCCL instead of being skipped
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29517')
        os.environ.setdefault('RANK', str(rank))
        os.environ.setdefault('WORLD_SIZE', str(world))
        dist.init_process_group('nccl', device_id=dev)
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)                        # RCCL is really connecting `world` ranks
        ranks_seen = int(round(float(ones.item())))
    dist_on = world > 1 or force_rccl

    if args.dtype == 'bf16':
        # bf16 STORAGE mode (renet_gemm_bf16s); RENET_BF16_STORAGE=0 keeps fp32 tensors and rounds inside the GEMMs.
        # Selected through the environment BEFORE the library wrapper is imported (it reads RENET_GEMM once): the
        # benchmark does not reach into the module's state
        os.environ['RENET_GEMM'] = 'bf16' if os.environ.get('RENET_BF16_STORAGE') == '0' else 'bf16s'
    import renet_hip as K
    K.lib()
    import model as M
    import parallel
    import preprocess as P
    import synth

    # ---- workload ------------------------------------------------------------------------------
    quads, num_ent, num_rels, unit = synth.make_stream(args.shape, seed=999)
    graph_dict = P.build_graph_dict(quads, num_rels)
    hist_s = P.HistoryIndex(quads, 's', history_len=args.seq_len)
    hist_o = P.HistoryIndex(quads, 'o', history_len=args.seq_len)
    np.random.seed(999)
    torch.manual_seed(999)
    net = M.RENet(num_ent, args.hidden, num_rels, dropout=args.dropout, seq_len=args.seq_len, num_k=1000)
    gen = torch.Generator().manual_seed(7)
    net.global_emb = {int(t): torch.randn(1, 1, args.hidden, generator=gen) * 0.1 for t in graph_dict}
    net.to(dev)
    net.train()
    # train.py:61,140-142: Adam(lr 1e-3, wd 1e-5) + clip_grad_norm_(1.0) + zero_grad, fused on flat buffers
    opt = parallel.HipAdam(net, lr=1e-3, weight_decay=1e-5, max_norm=1.0)
    flat = opt.grads
    perm = np.random.RandomState(999).permutation(len(quads))

    if args.no_pair:
        args.passes = 'serial'

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    import ops as ops_mod_

    class Mode(parallel.ScalingMode):
        """One scaling mode of the data-parallel step (parallel.ScalingMode: which quadruples a rank takes, how gradients
        combine) bound to this run's network, data and optimizer."""

        def __init__(self, scaling):
            super().__init__(scaling, args.batch, world, passes=args.passes)

        def prepare(self, step):
            idx = self.indices(perm, step, rank)
            b = quads[idx]
            if self.passes == 'merged':
                both = net.prepare_both(b, hist_s.take(idx), hist_o.take(idx), graph_dict, shard=self.shard(rank))
                if both is not None:
                    return (both,)
                assert not self.exact, 'exact scaling needs histories on both sides of the batch'
            return (net.prepare(b, hist_s.take(idx), graph_dict, subject=True),
                    net.prepare(b, hist_o.take(idx), graph_dict, subject=False))

        def step_loss(self, *preps):
            if len(preps) == 1:              # same arithmetic per row; every kernel sees the rows of both passes
                return net.loss_prepared_both(preps[0])
            if self.passes == 'serial':
                return net.loss_prepared(preps[0]) + net.loss_prepared(preps[1])
            return net.loss_prepared_pair(*preps)       # the four GRU recurrences of the two passes share one launch

        def train_step(self, *preps):
            with opt.step_scope(head_passes=1 if len(preps) == 1 else 2, average=self.average):
                loss = self.step_loss(*preps)
                loss.backward()
                opt.step()                   # gradient all-reduce (N>1) -> clip -> Adam -> zero_grad
            return loss

        def run(self, timer=None, prepared=None):
            """W untimed + K timed steps on device-resident batches -> (elapsed seconds [max over ranks], last
            loss, host build ms per step, the prepared batches).  timer: a K.KernelTimer -- HIP events around every
            C-ABI launch (one stream, serial launches); `value` comes from a run WITHOUT one."""
            n_total = args.warmup + args.steps
            host_ms = None
            if prepared is None:
                t0 = time.time()
                prepared = [self.prepare(k) for k in range(n_total)]
                torch.cuda.synchronize()
                host_ms = (time.time() - t0) * 1e3 / max(n_total, 1)
            for k in range(args.warmup):
                self.train_step(*prepared[k])
            assert flat.check_views(), 'param.grad views were replaced'
            if timer is not None:
                K.set_timer(timer, glue=True)            # every C-ABI launch of the step is timed (classes 'glue:*')
            sync_all()
            t0 = time.perf_counter()
            for k in range(args.warmup, n_total):
                loss = self.train_step(*prepared[k])
            sync_all()
            elapsed = time.perf_counter() - t0
            K.set_timer(None)
            last = float(loss.item())
            if world > 1:
                tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                elapsed = float(tmax.item())
            return elapsed, last, host_ms, prepared

        def per_step_pass(self, prepared, n_steps):
            """`n_steps` more steps over the prepared batches (cyclically) with ONE HIP event between consecutive steps ->
            the per-step device times in ms (median / spread: a K-step wall-clock window of 55 ms decides nothing).
            Product configuration (no per-kernel events); parameters keep training."""
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_steps + 1)]
            sync_all()
            evs[0].record()
            for k in range(n_steps):
                self.train_step(*prepared[args.warmup + k % args.steps])
                evs[k + 1].record()
            sync_all()
            return [evs[k].elapsed_time(evs[k + 1]) for k in range(n_steps)]

        def enqueue_pass(self, prepared, n_steps=16):
            """Host time to ENQUEUE a step (no synchronisation inside), over a window short enough that the launch queues
            never fill: once the host is hundreds of launches ahead of the GPU it blocks on queue space and the figure
            degenerates into the device time per step."""
            sync_all()
            t0 = time.perf_counter()
            for k in range(n_steps):
                self.train_step(*prepared[args.warmup + k % args.steps])
            dt = time.perf_counter() - t0
            sync_all()
            return dt * 1e3 / n_steps

    mode = Mode(args.scaling)
    args.passes = mode.passes
    exact_split, rank_batch = mode.exact, mode.rank_batch
    prepare, step_loss, train_step = mode.prepare, mode.step_loss, mode.train_step
    n_total = args.warmup + args.steps
    # the timed region of `value`: the product configuration (side streams on, no per-kernel events) ...
    elapsed, last_loss, host_build_ms, prepared = mode.run()
    value = mode.global_batch * args.steps / elapsed
    # the same step measured two more ways (product configuration, same prepared batches): per-step HIP events over >= 100
    # steps whatever --steps says (median and spread), and the host's enqueue time over a 16-step window
    median_block = host_enqueue_ms = None
    if not args.plain:                       # (--plain: the W + K steps of `value` alone, for a profiler)
        per_step = mode.per_step_pass(prepared, max(100, args.steps) if not args.child else max(20, args.steps))
        per_step_sorted = sorted(per_step)
        ms_median = per_step_sorted[len(per_step_sorted) // 2]
        median_block = {'value': mode.global_batch * 1e3 / ms_median, 'ms_per_step': ms_median, 'steps': len(per_step),
                        'p10_ms': per_step_sorted[len(per_step_sorted) // 10],
                        'p90_ms': per_step_sorted[(len(per_step_sorted) * 9) // 10],
                        'what': 'median of per-step HIP-event intervals (one event between consecutive steps, product '
                                'configuration); `value` is the contract\'s K-step wall-clock window'}
        host_enqueue_ms = mode.enqueue_pass(prepared)
    elif os.environ.get('RENET_BENCH_PLAIN_ENQUEUE') == '1':
        host_enqueue_ms = mode.enqueue_pass(prepared)
    import step_plan as step_plan_mod
    plan_entries = sum(step_plan_mod.StepFn.last_launches) if (step_plan_mod.ENABLED and sum(step_plan_mod.StepFn.last_launches)) else None
    # ... then the SAME K steps once more with HIP events around every C-ABI launch: the per-class kernel table, the
    # roofline's average launch durations and `kernel_only` come from this second pass (its wall time is reported too)
    timer = K.KernelTimer()
    elapsed_timed = None
    if not args.plain:
        elapsed_timed, _, _, _ = mode.run(timer, prepared)

    # ---- N > 1: the other scaling modes of the same step, so that the line is explicit about global batch size ----
    companions = {}
    if world > 1 and args.companions:
        for sc in ('weak', 'strong', 'exact'):
            if sc == args.scaling:
                continue
            m2 = Mode(sc)
            el2, loss2, _, _ = m2.run()
            companions['scaling_' + sc] = {'value': m2.global_batch * args.steps / el2,
                                           'ms_per_step': el2 * 1e3 / args.steps, 'global_batch': m2.global_batch,
                                           'batch_per_gpu': m2.rank_batch, 'last_loss': loss2}

    # ---- encoder-only companion: RGCN x2 + sequence assembly + GRU x2, forward + backward (SURVEY 8d asks for the
    # encoder-only device rate next to the full step; the metric is NAMED "RGCN+GRU encoder triples/s").  The heads
    # are replaced by fixed random cotangents on h_n / q_n; no clip / Adam; gradients are zeroed outside the timing.
    encoder_only = None
    if args.enc_steps > 0 and world == 1 and len(prepared[args.warmup]) == 1:
        cot = torch.randn(2, 2 * rank_batch, args.hidden, device=dev)

        def enc_step(prep):
            x, xr = net.aggregator.encode(prep.g, net.ent_embeds, net.rel_embeds, reverse=False)
            s_h, s_q = ops_mod.dual_gru(x, xr, net.encoder, net.encoder_r, prep.step_off, prep.b)
            ((s_h[0] * cot[0, :prep.b]).sum() + (s_q[0] * cot[1, :prep.b]).sum()).backward()
        import ops as ops_mod
        for k in range(3):
            enc_step(prepared[k][0])
        sync_all()
        t0 = time.perf_counter()
        for k in range(args.enc_steps):
            enc_step(prepared[args.warmup + (k % args.steps)][0])
        sync_all()
        dt = time.perf_counter() - t0
        flat.zero()
        encoder_only = {'value': rank_batch * args.enc_steps / dt, 'unit': 'triples/s', 'ms_per_step': dt * 1e3 / args.enc_steps,
                        'steps': args.enc_steps,
                        'what': 'RGCN x2 + sequence assembly + GRU x2 (both encoders), both directions, forward + '
                                'backward incl. their parameter gradients; no score heads, clip or Adam'}

    # ---- the reference's loop body through the API it calls (train.py:133-142), unmodified statement by statement:
    # nested history lists as train.py holds them in memory (unpickled once, sliced per batch by utils.make_batch2), the
    # TWO model() calls of a batch, loss.backward(), torch's clip_grad_norm_, torch.optim.Adam, zero_grad, loss.item().
    # `fuse_directions` (the two calls share one merged pass) and the device batch builder are the package's switches for
    # this API; everything else is what a user of the reference runs.  End to end: list flattening, upload, batch build,
    # step and the per-step host sync are all inside the timed region.
    value_list_api = None
    if args.list_steps > 0 and world == 1:
        model = M.RENet(num_ent, args.hidden, num_rels, dropout=args.dropout, seq_len=args.seq_len, num_k=1000)
        model.global_emb = net.global_emb
        model.to(dev)
        model.train()
        model.fuse_directions = True
        optimizer = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
        n_list = args.list_steps + 5
        batches = []
        for k in range(n_list):
            idx = parallel.shard_indices(perm, n_total + 5000 + k, 0, 1, args.batch)
            s_hist, s_hist_t = hist_s.to_lists(idx)
            o_hist, o_hist_t = hist_o.to_lists(idx)
            batches.append((quads[idx], s_hist, s_hist_t, o_hist, o_hist_t))
        loss_epoch = 0
        t0 = None
        marks = []
        for k, (batch_data, s_hist, s_hist_t, o_hist, o_hist_t) in enumerate(batches):
            if k == 5:                                   # (5 warm-up steps: builder capacities, allocator, first-use uploads)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            marks.append(time.perf_counter())            # (every step ends in loss.item(): a host sync)
            batch_data = torch.from_numpy(batch_data).long()
            batch_data = batch_data.cuda()
            loss_s = model(batch_data, (s_hist, s_hist_t), (o_hist, o_hist_t), graph_dict, subject=True)
            loss_o = model(batch_data, (s_hist, s_hist_t), (o_hist, o_hist_t), graph_dict, subject=False)
            loss = loss_s + loss_o
            loss.backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)  # clip gradients
            optimizer.step()
            optimizer.zero_grad()
            loss_epoch += loss.item()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        marks.append(time.perf_counter())
        per = sorted((b_ - a_) * 1e3 for a_, b_ in zip(marks[5:-1], marks[6:]))
        value_list_api = {'value': args.batch * args.list_steps / dt, 'unit': 'triples/s',
                          'ms_per_step': dt * 1e3 / args.list_steps, 'steps': args.list_steps,
                          'ms_per_step_median': per[len(per) // 2], 'ms_per_step_max': per[-1],
                          'c_launch_list': bool(step_plan_mod.ENABLED),
                          'what': "the reference's loop body (train.py:133-142) over this package's RENet: nested lists, "
                                  'two model() calls (fuse_directions on), loss.backward(), clip_grad_norm_, '
                                  'torch.optim.Adam, zero_grad, loss.item() -- end to end incl. list flattening, upload, '
                                  'device batch build and the per-step host sync'}
        del model, optimizer, batches

    # ---- end-to-end rates with the host builder in the loop (extras, never `value`) ----------------
    #  e2e_inline : builder on the training thread (one batch at a time)
    #  e2e_value  : builder in forked worker processes (pipeline.BatchPrefetcher), uploads on this thread
    #  e2e_threads8 : builder in 8 threads of THIS process (no fork; the native passes and numpy release the GIL)
    e2e = e2e_inline = e2e_threads = e2e_device_builder = device_build_ms = None
    e2e_workers = 0
    if args.e2e_steps > 0 and world == 1:      # single-GPU extra; multi-GPU runs time the device path only
        sync_all()
        t0 = time.perf_counter()
        for k in range(args.e2e_steps):
            train_step(*prepare(n_total + k))
        sync_all()
        e2e_inline = args.batch * world * args.e2e_steps / (time.perf_counter() - t0)
        # the same loop with the DEVICE batch builder (csrc/builder.hip): no worker processes, no builder threads; the
        # builder kernels of batch k + 1 run on a side stream while step k runs, the training thread only launches
        if args.passes == 'merged':
            import gpu_builder
            dstore = gpu_builder.DeviceStore(quads, hist_s, hist_o, graph_dict, net.global_emb, num_ent, num_rels, dev)
            side = torch.cuda.Stream()

            def dev_pending(step):
                return net.prepare_both_device(parallel.shard_indices(perm, step, rank, world, rank_batch), dstore,
                                               stream=side)

            def dev_finish(pend, step):
                prep_ = net.finish_prepare_device(pend)
                while prep_ is None:                             # a capacity grew: rebuild (first batches only)
                    prep_ = net.finish_prepare_device(dev_pending(step))
                return prep_
            n_dev = max(30, 6 * args.e2e_steps)
            base = n_total + 2000
            pend = dev_pending(base)
            for k in range(3):                                   # warm-up (capacities, allocator)
                nxt = dev_pending(base + k + 1)
                train_step(dev_finish(pend, base + k))
                pend = nxt
            sync_all()
            t0 = time.perf_counter()
            for k in range(3, 3 + n_dev):
                nxt = dev_pending(base + k + 1)
                train_step(dev_finish(pend, base + k))
                pend = nxt
            sync_all()
            e2e_device_builder = args.batch * world * n_dev / (time.perf_counter() - t0)
            dev_finish(pend, base + 3 + n_dev)
            # the builder alone on an idle GPU
            sync_all()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            pends = [net.prepare_both_device(parallel.shard_indices(perm, base + 100 + k, rank, world, rank_batch), dstore)
                     for k in range(10)]
            ev1.record()
            torch.cuda.synchronize()
            device_build_ms = ev0.elapsed_time(ev1) / 10
            del pends
        import pipeline

        def host_step(step):
            idx = parallel.shard_indices(perm, step, rank, world, rank_batch)
            b = quads[idx]
            if args.passes == 'merged':
                both = net.host_batch_both(b, hist_s.take(idx), hist_o.take(idx), graph_dict)
                if both is not None:
                    return (both,)
            return (net.host_batch(b, hist_s.take(idx), graph_dict, subject=True),
                    net.host_batch(b, hist_o.take(idx), graph_dict, subject=False))
        n_pipe = max(40, 8 * args.e2e_steps)
        e2e_workers = pipeline.worker_budget(world)
        pf = pipeline.BatchPrefetcher(host_step, range(n_total + 100, n_total + 100 + n_pipe), e2e_workers)
        up = pipeline.Uploader(net)                              # H2D on a copy stream, off the compute queue
        it = iter(pf)
        first = [next(it) for _ in range(4)]                    # let the workers fill the pipe
        for hbs in first:
            train_step(*[up(h) for h in hbs])
        sync_all()
        t0 = time.perf_counter()
        done = 0
        for hbs in it:
            train_step(*[up(h) for h in hbs])
            done += 1
        sync_all()
        e2e = args.batch * world * done / (time.perf_counter() - t0)
        # the same pipeline with 8 builder THREADS of this process instead of forked workers
        pf = pipeline.BatchPrefetcher(host_step, range(n_total + 1000, n_total + 1000 + n_pipe), 8, threads=True)
        it = iter(pf)
        for hbs in [next(it) for _ in range(4)]:
            train_step(*[up(h) for h in hbs])
        sync_all()
        t0 = time.perf_counter()
        done = 0
        for hbs in it:
            train_step(*[up(h) for h in hbs])
            done += 1
        sync_all()
        e2e_threads = args.batch * world * done / (time.perf_counter() - t0)

    if rank != 0:
        if dist_on:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel class (HIP events recorded on the launch stream) ---------
    # `traffic`: HBM bytes per launch.  Hardware counters cannot be read from inside the process: they come from
    # the rocprofv3 --pmc passes of this same command (FETCH_SIZE / WRITE_SIZE in separate runs, FETCH doubled
    # for 16 B/lane reads per MI355X_MICROARCH.md; tools/pmc_traffic.py), committed under profiles/ -- the newest
    # file collected in THIS GEMM mode is used, and only for the workload it was collected on; otherwise null.
    pmc, pmc_file = {}, None
    legacy_mode = {'r02_pmc_traffic.json': 'bf16x6', 'r03_pmc_traffic.json': 'f16x3'}
    for name in sorted(os.listdir(os.path.join(ROOT, 'profiles')), reverse=True) \
            if os.path.isdir(os.path.join(ROOT, 'profiles')) else []:
        if '_pmc_traffic' in name and name.endswith('.json'):
            try:
                with open(os.path.join(ROOT, 'profiles', name)) as f:
                    rec = json.load(f)
            except (OSError, ValueError):
                continue
            if rec.get('gemm_mode', legacy_mode.get(name)) == K.GEMM_MODE:
                pmc, pmc_file = rec.get('classes', {}), 'profiles/' + name
                break
    same_workload = (args.shape == 'ICEWS18' and args.batch == 1024 and args.hidden == 200 and args.seq_len == 10)

    def traffic_of(name):
        return pmc[name]['hbm_bytes_per_launch'] if (same_workload and name in pmc) else None
    stats = timer.summary()
    g0 = prepared[args.warmup][0].g.host
    kernels = {}
    for name, st in stats.items():
        ent = {'calls_per_step': st['calls'] / args.steps, 'avg_us': st['ms'] * 1e3 / max(st['calls'], 1),
               'ms_per_step': st['ms'] / args.steps}
        if st['flops']:
            ent['tflops'] = st['flops'] / (st['ms'] * 1e-3) / 1e12
        if st['bytes']:
            ent['gbs'] = st['bytes'] / (st['ms'] * 1e-3) / 1e9
        kernels[name] = ent
    # per-shape view of the GEMM class (row counts that vary with the batch graph are rounded to thousands)
    merged = {}
    for tag, o in timer.by_tag('gemm_f32').items():
        if tag is None:
            continue
        key = tuple(tag[:2]) + tuple(v if v <= 2048 or v == num_ent else int(round(v, -3)) for v in tag[2:5]) + (tag[5],)
        mo = merged.setdefault(key, {'calls': 0, 'ms': 0.0, 'flops': 0.0})
        for f in mo:
            mo[f] += o[f]
    gemm_shapes = [{'ta_tb_m_n_k_split': list(key), 'calls_per_step': round(o['calls'] / args.steps, 2),
                    'avg_us': round(o['ms'] * 1e3 / o['calls'], 2),
                    'tflops': round(o['flops'] / (o['ms'] * 1e-3) / 1e12, 1)}
                   for key, o in sorted(merged.items(), key=lambda kv: -kv[1]['ms'])]
    named = {n: st for n, st in stats.items() if not n.startswith('glue:')}
    dom = max(named, key=lambda n: named[n]['ms']) if named else None
    # matrix-pipe ceilings for ALGORITHMIC fp32 flops (2MNK) per GEMM mode: f32 = the f32-input MFMA peak; bf16x6 = six
    # bf16 products per fp32 product, dense bf16 peak / 6; f16x3 = three f16 products, / 3; bf16 / bf16s = one product
    gemm_peak = {'f16x3': MFMA_BF16_PEAK_TF / 3.0, 'bf16x6': MFMA_BF16_PEAK_TF / 6.0, 'bf16': MFMA_BF16_PEAK_TF,
                 'bf16s': MFMA_BF16_PEAK_TF}.get(K.GEMM_MODE, MFMA_F32_PEAK_TF)
    gemm_note = {'f16x3': 'algorithmic fp32 TFLOP/s; peak = f16 dense 2500/3 (three f16 MFMA products per fp32 product)',
                 'bf16x6': 'algorithmic fp32 TFLOP/s; peak = bf16 dense 2500/6 (six bf16 MFMA products per fp32 '
                           'product)', 'bf16': 'bf16 dense MFMA peak',
                 'bf16s': 'bf16 dense MFMA peak (bf16 operands in HBM)'}.get(K.GEMM_MODE, 'f32-input MFMA peak')
    roofline = None
    if dom:
        st = stats[dom]
        if st['flops']:
            ach = st['flops'] / (st['ms'] * 1e-3) / 1e12
            operand_bytes = None
            if dom == 'gemm_f32' and merged:
                # operands + outputs of the class per launch (fp32 elements; what `traffic` is to be read against)
                tot = sum(o['calls'] * 4.0 * (k_[2] * k_[4] + k_[3] * k_[4] + k_[2] * k_[3]) for k_, o in merged.items())
                operand_bytes = tot / max(1, sum(o['calls'] for o in merged.values()))
            roofline = {'kernel': dom, 'bound': 'mfma', 'achieved': ach, 'peak': gemm_peak,
                        'unit': 'TFLOP/s', 'frac': ach / gemm_peak, 'traffic': traffic_of(dom),
                        'operand_bytes_per_launch': operand_bytes, 'avg_us': st['ms'] * 1e3 / st['calls'],
                        'calls_per_step': st['calls'] / args.steps, 'gemm_mode': K.GEMM_MODE, 'note': gemm_note}
        else:
            ach = st['bytes'] / (st['ms'] * 1e-3) / 1e9
            roofline = {'kernel': dom, 'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                        'frac': ach / HBM_PEAK_GBS, 'traffic': traffic_of(dom)}
    # the north-star kernel: one entry per launch class of the step (distinct kernel names in a rocprof trace).
    # `achieved` uses SURVEY 8d's bytes incl. the fused self-loop addend row where the launch has one (the forward
    # launches; the backward launches run without an addend); `frac_strict` never counts it.
    gather = {}
    d = args.hidden
    for name in ('rgcn_gather_fwd_full', 'rgcn_gather_bwdh_full', 'rgcn_gather_fwd_pruned', 'rgcn_gather_bwdh_pruned'):
        if name in stats:
            st = stats[name]
            ach = st['bytes'] / (st['ms'] * 1e-3) / 1e9
            strict = sum(t * o['calls'] for t, o in timer.by_tag(name).items()) / (st['ms'] * 1e-3) / 1e9
            gather[name] = {'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                            'frac': ach / HBM_PEAK_GBS, 'frac_strict': strict / HBM_PEAK_GBS,
                            'traffic': traffic_of(name), 'avg_us': st['ms'] * 1e3 / st['calls'],
                            'calls_per_step': st['calls'] / args.steps,
                            'algorithmic_bytes_per_launch': st['bytes'] / st['calls']}
    gru = None
    if 'gru_recurrence' in stats:
        st = stats['gru_recurrence']
        ach = st['flops'] / (st['ms'] * 1e-3) / 1e12
        # the recurrent products run bf16x6 in both split modes, exact fp32 MFMAs in f32 mode, one bf16 product in the
        # bf16 modes (gru.hip)
        peak = {'bf16x6': MFMA_BF16_PEAK_TF / 6.0, 'f16x3': MFMA_BF16_PEAK_TF / 6.0, 'bf16': MFMA_BF16_PEAK_TF,
                'bf16s': MFMA_BF16_PEAK_TF}.get(K.GEMM_MODE, MFMA_F32_PEAK_TF)
        gru = {'kernel': 'gru_fwd/bwd recurrence (both encoders per launch)', 'bound': 'mfma', 'achieved': ach,
               'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak, 'avg_us': st['ms'] * 1e3 / st['calls'],
               'calls_per_step': st['calls'] / args.steps,
               'note': 'products: %s; in practice bound by the W_hh stream L2 -> LDS (DESIGN 4c)' % (
                   {'bf16x6': 'bf16x6', 'f16x3': 'bf16x6', 'bf16': 'one bf16 plane', 'bf16s': 'one bf16 plane'}.get(
                       K.GEMM_MODE, 'exact fp32'))}

    # ---- CPU baseline: the oracle (restated reference path) on this box's host cores; the same oracle steps give
    # the parity check: HIP eval-mode loss vs oracle loss on identical batches and (current) parameters ----------
    cpu = parity = None
    if args.cpu_steps > 0 and world == 1:
        cpu, oracle_losses, cpu_steps_idx, oracle_grad = cpu_baseline(args, quads, num_ent, num_rels, hist_s, hist_o,
                                                                      net, perm)
        net.eval()
        hip_losses = []
        with torch.no_grad():
            for k in cpu_steps_idx:
                hip_losses.append(float(step_loss(*prepare(k)).item()))      # the path that was timed
        # gradient of the first checked batch: the HIP backward pass (eval-mode masks, no optimizer step) against the
        # oracle's autograd, in the flat parameter layout -- whole-vector relative error, norms, and 4096 seeded samples
        flat.zero()
        step_loss(*prepare(cpu_steps_idx[0])).backward()
        g_hip = flat.flat.detach().cpu().double().numpy()
        flat.zero()
        net.train()
        g_ref = np.zeros_like(g_hip)
        names = {id(p_): n_ for n_, p_ in net.named_parameters()}
        for p_, off in zip(flat.params, flat.offsets):
            g_ref[off:off + p_.numel()] = oracle_grad[names[id(p_)]].reshape(-1)
        samp = np.random.RandomState(4096).randint(0, g_ref.size, size=4096)
        rel = [abs(a - b_) / abs(b_) for a, b_ in zip(hip_losses, oracle_losses)]
        parity = {'hip_loss': hip_losses[0], 'oracle_loss': oracle_losses[0], 'rel_err': max(rel),
                  'batches': len(rel), 'mode': 'eval (dropout off), parameters after the timed steps',
                  'tolerance': 2e-3 if args.dtype == 'bf16' else 2e-4,
                  'grad_rel_err': float(np.linalg.norm(g_hip - g_ref) / np.linalg.norm(g_ref)),
                  'grad_norm_hip': float(np.linalg.norm(g_hip)), 'grad_norm_oracle': float(np.linalg.norm(g_ref)),
                  'grad_sampled_max_err_over_max': float(np.abs(g_hip[samp] - g_ref[samp]).max() / np.abs(g_ref).max()),
                  'grad_tolerance': 5e-2 if args.dtype == 'bf16' else 2e-3}

    # ---- companions (child processes of this command): the other fp32-class GEMM modes as FULL records (value, ms per
    # step, kernels, gemm_shapes, roofline with that mode's own ceiling and PMC file), and the other BASELINE.json
    # configs (configs[2..4]) with their parity against the oracle step
    def child(extra_args, env_extra, steps, cpu_steps=0):
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), '--steps', str(steps), '--warmup', str(args.warmup),
               '--batch', str(args.batch), '--dropout', str(args.dropout), '--cpu-steps', str(cpu_steps),
               '--cpu-warmup', '0', '--e2e-steps', '0', '--list-steps', '0', '--f32-steps', '0', '--enc-steps', '0', '--other-steps', '0',
               '--passes', args.passes, '--child', '1'] + extra_args
        r = subprocess.run(cmd, env=dict(os.environ, **env_extra), capture_output=True, text=True)
        try:
            j = json.loads(r.stdout.strip().splitlines()[-1])
        except (ValueError, IndexError):
            return {'error': (r.stderr or r.stdout)[-300:]}
        keep = ('value', 'ms_per_step', 'steps', 'dtype', 'gemm_mode', 'precision', 'last_loss', 'roofline',
                'roofline_rgcn_gather', 'roofline_gru', 'kernels', 'gemm_shapes', 'kernel_only', 'parity',
                'pmc_source')
        rec = {k_: j.get(k_) for k_ in keep}
        rec['workload'] = j.get('config', {}).get('workload')
        return rec

    same = ['--shape', args.shape, '--hidden', str(args.hidden), '--seq-len', str(args.seq_len)]
    modes = {}
    if args.f32_steps > 0 and world == 1 and not args.child and K.GEMM_MODE in ('bf16x6', 'f16x3', 'f32'):
        for md in ('bf16x6', 'f16x3', 'f32'):
            if md != K.GEMM_MODE:
                modes[md] = child(same, {'RENET_GEMM': md}, args.f32_steps)
    other_configs = None
    if args.other_steps > 0 and world == 1 and not args.child and same_workload and args.dtype == 'f32':
        other_configs = {
            'wiki_d200': child(['--shape', 'WIKI'], {}, args.other_steps, cpu_steps=1),
            'gdelt_d200': child(['--shape', 'GDELT'], {}, args.other_steps, cpu_steps=1),
            'yago_d400_l15_bf16': child(['--shape', 'YAGO', '--hidden', '400', '--seq-len', '15', '--dtype', 'bf16'], {},
                                        args.other_steps, cpu_steps=1),
        }
        for rec in other_configs.values():          # the per-shape tables stay in the child's own run: keep the line short
            rec.pop('gemm_shapes', None)
            rec.pop('kernels', None)

    ko = sum(e_['ms_per_step'] for e_ in kernels.values())
    glue_ms = sum(e_['ms_per_step'] for n_, e_ in kernels.items() if n_.startswith('glue:'))
    precision = {'f16x3': 'fp32 storage; GEMM operands split into two tensor-scaled binary16 planes (22 significant '
                          'bits), three f16 MFMA products, fp32 accumulation (DESIGN 3f)',
                 'bf16x6': 'fp32 storage; GEMM operands split into three bf16 planes (all 24 significand bits), six bf16 '
                           'MFMA products, fp32 accumulation: fp32-class',
                 'f32': 'fp32 storage, exact fp32 MFMA products (v_mfma_f32_32x32x2_f32)',
                 'bf16s': 'bf16 storage of the GEMM operands, fp32 accumulation',
                 'bf16': 'fp32 storage, operands rounded to bf16 in the GEMM loaders'}.get(K.GEMM_MODE)
    out = {
        'metric': 'RGCN+GRU encoder triples/s at bs=%d n_hidden=%d (full training step, both directions)'
                  % (args.batch, args.hidden),
        'value': value, 'unit': 'triples/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': elapsed * 1e3 / args.steps, 'higher_is_better': True, 'scaling': mode.reported_scaling,
        'scaling_mode': args.scaling, 'global_batch': mode.global_batch, 'rccl_ranks_seen': ranks_seen,
        'vs_baseline': None,
        # the arithmetic type the path computes in: fp32 tensors everywhere ('f32'); --dtype bf16 = bf16 operand storage
        'dtype': args.dtype, 'data': 'synthetic', 'gemm_mode': K.GEMM_MODE, 'precision': precision,
        'passes': args.passes,
        'config': {'workload': '%s-shaped synthetic stream (seed 999), n_hidden=%d, seq_len=%d, batch=%d per GPU, '
                               'dropout=%.2f, fwd+bwd both directions + clip + Adam' %
                               (args.shape, args.hidden, args.seq_len, rank_batch, args.dropout),
                   'num_entities': num_ent, 'num_relations': num_rels, 'parallelism': 'dp%d' % world,
                   'batch_graph': {'nodes': int(g0.N), 'edges': int(g0.E), 'history_steps': int(g0.S),
                                   'nonempty': int(g0.nnz)}},
        'roofline': roofline, 'roofline_rgcn_gather': gather, 'roofline_gru': gru, 'parity': parity,
        'encoder_only': encoder_only,
        'kernel_only': {'ms_per_step': ko, 'value': rank_batch / max(ko * 1e-3, 1e-12), 'glue_ms_per_step': glue_ms,
                        'timed_pass_ms_per_step': elapsed_timed * 1e3 / args.steps if elapsed_timed else None,
                        'what': 'sum of the HIP-event durations of EVERY C-ABI launch of a step (named classes + the '
                                "small kernels as 'glue:*'), from a second pass over the same steps with events on "
                                '(serial launches, no side stream); torch-native launches (a dot, a few fills / adds) are '
                                'the only ones outside it'},
        'value_bf16x6': modes.get('bf16x6'), 'value_f16x3': modes.get('f16x3'), 'value_exact_f32': modes.get('f32'),
        'other_configs': other_configs,
        'pmc_source': pmc_file,
        'traffic_source': ('%s: rocprofv3 --pmc passes of this command (tools/pmc_traffic.py), NOT measured in this run' % pmc_file) if pmc_file else None, 'kernels': kernels, 'gemm_shapes': gemm_shapes, 'cpu_baseline': cpu,
        'launches_per_step': sum(e_['calls_per_step'] for e_ in kernels.values()) or None,     # C-ABI launches (autograd path, timed pass)
        'plan_entries_per_step': plan_entries,
        'value_list_api': value_list_api,             # launches the C launch list issues per step (csrc/step.cpp: forward + backward)
        'value_median': median_block,
        'host_enqueue_ms_per_step': host_enqueue_ms,       # host time to enqueue a step of the `value` run; close to
                                                           # ms_per_step = the host, not the GPU, paces the step
        'host_build_ms': host_build_ms, 'e2e_value': e2e, 'e2e_workers': e2e_workers, 'e2e_inline': e2e_inline, 'e2e_threads8': e2e_threads,
        'e2e_device_builder': e2e_device_builder, 'device_build_ms': device_build_ms,
        'last_loss': last_loss,
    }
    out.update(companions)
    if world > 1 and not companions:
        out['multi_gpu_note'] = 'companion scaling modes skipped (--companions 0)'
    if args.child:
        print(json.dumps(out))            # a companion's full record, read by the parent process only
    else:
        print(compact_line(out, write_detail(out)))
    if dist_on:
        dist.destroy_process_group()


def cpu_baseline(args, quads, num_ent, num_rels, hist_s, hist_o, net, perm):
    """Times the oracle (oracle/renet_oracle.py: the reference's algorithm restated on torch-CPU; the reference itself
    -- Python + DGL -- cannot travel to this box) on a bounded sample of the SAME workload, in the SAME mode as the GPU
    step: `cpu_steps` TRAINING steps (forward both directions with the five dropout sites at --dropout, backward) at the
    same batch size after `cpu_warmup` untimed ones, per-stage medians.  The parity check needs deterministic losses: a few
    of the same batches are evaluated once more in eval mode (not timed).
    Returns (record, eval-mode oracle losses, their step indices, the eval-mode gradient of the first).  Reported baseline,
    not a target."""
    from oracle import renet_oracle as O
    import parallel
    threads = args.cpu_threads or min(32, os.cpu_count() or 1)
    old_threads = torch.get_num_threads()
    torch.set_num_threads(threads)
    params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    ge = {t: v.view(-1).cpu() for t, v in net.global_emb.items()}
    ogd = O.build_graph_dict(quads, num_rels)
    warm = max(0, args.cpu_warmup)

    def one_step(step, dropout, tm=None):
        idx = parallel.shard_indices(perm, step, 0, 1, args.batch)
        hs, hst = hist_s.to_lists(idx)
        ho, hot = hist_o.to_lists(idx)
        t0 = time.perf_counter()
        loss = O.renet_forward_loss(params, quads[idx], hs, hst, ogd, ge, num_rels, args.seq_len, subject=True,
                                    timer=tm, dropout=dropout) + \
            O.renet_forward_loss(params, quads[idx], ho, hot, ogd, ge, num_rels, args.seq_len, subject=False, timer=tm,
                                 dropout=dropout)
        if tm is not None:
            tm.reset()
        loss.backward()
        if tm is not None:
            tm.mark('backward')
        return loss, time.perf_counter() - t0

    times, stages = [], []
    torch.manual_seed(1234)
    for k in range(args.cpu_steps + warm):                 # the TIMED sample: train mode, like the GPU step
        tm = O.StageTimer()
        loss, dt = one_step(1000 + k, args.dropout, tm)
        for p in params.values():
            p.grad = None
        if k >= warm:
            times.append(dt)
            stages.append(dict(tm.t))
    losses, steps_idx, first_grad = [], [], None
    for k in range(min(3, args.cpu_steps)):                # the checker: eval mode (deterministic), not timed
        step = 1000 + warm + k
        loss, _ = one_step(step, 0.0)
        if first_grad is None:                             # for bench `parity.grad_rel_err`
            first_grad = {n: (p.grad.detach().double().numpy() if p.grad is not None else np.zeros(tuple(p.shape)))
                          for n, p in params.items()}
        for p in params.values():
            p.grad = None
        losses.append(float(loss.item()))
        steps_idx.append(step)
    torch.set_num_threads(old_threads)
    t = float(np.median(times))
    stage_ms = {k: float(np.median([s_[k] for s_ in stages])) * 1e3 for k in stages[0]}
    rec = {'value': args.batch / t, 'unit': 'triples/s', 'cores': int(threads), 'kind': 'port',
           'host_cpus': os.cpu_count(), 'ms_per_step': t * 1e3, 'stage_ms_median': stage_ms, 'mode': 'train',
           'sample': '%d TRAINING steps (fwd both directions with dropout %.2f at the five nn.Dropout sites + bwd) of batch '
                     '%d after %d warm-ups, oracle/renet_oracle.py (restated reference; the Python + DGL reference cannot '
                     'travel to the GPU box) on torch-CPU with %d threads; medians'
                     % (len(times), args.dropout, args.batch, warm, threads)}
    return rec, losses, steps_idx, first_grad


if __name__ == '__main__':
    main()
