"""Benchmark of the PS-VAE train step on MI355X (one process per GPU).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = collate a batch of synthetic ZINC-shaped molecules (already packed in HBM) on the
GPU, forward, backward, gradient all-reduce (N > 1), global-norm clip, Adam.  Workload:
BASELINE.json configs[2] ("synthetic ZINC-shaped graphs, batch 4096") per GPU, weak scaling.
Rank 0 prints ONE JSON line: metric / value / ms_per_step, the roofline of the dominant kernels from event
brackets on a few timed steps, sub-records for batch 32 and 10k-molecule generation, and a CPU baseline.
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E, spec
MFMA_F32_PEAK_TFLOPS = 157.3  # v_mfma_f32_32x32x2_f32, dense


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=4096, help='molecules per GPU per step')
    ap.add_argument('--dataset', type=int, default=0, help='synthetic molecules per rank (default 4x batch)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-sub-records', action='store_true', help='skip the batch-32 and 10k-generation sub-records')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend (nccl = RCCL; gloo for rehearsals)')
    ap.add_argument('--device', type=int, default=-1,
                    help='GPU index for every rank (rehearsal of N ranks on one GPU); default LOCAL_RANK')
    ap.add_argument('--gemm-shapes', action='store_true', help='per-shape GEMM timing table on stderr')
    ap.add_argument('--cpu-batch', type=int, default=256)
    ap.add_argument('--cpu-steps', type=int, default=30, help='CPU-baseline headline sample: about 8 s of host work')
    ap.add_argument('--cpu-full-steps', type=int, default=2,
                    help='CPU-baseline steps at the full batch of the headline workload (0: skip that leg)')
    return ap.parse_args()


REAL_DIR = os.path.join(ROOT, 'tests', 'golden', 'real')


def real_tokenizer(shape):
    from ps_vae_amd.utils.tokenizer import Tokenizer
    return Tokenizer(os.path.join(REAL_DIR, f'{shape}_vocab_300.txt'))


def make_dataset(num, seed, tok, device, shape='zinc'):
    import ps_vae_amd  # noqa: F401
    from ps_vae_amd.data.bpe_dataset import BPEMolDataset
    from ps_vae_amd.data.packed import PackedMolecules
    from ps_vae_amd.data.synthetic import synthetic_step1
    if shape.endswith('_real'):
        # real molecules: 2048 SMILES lines of the reference's data files (tests/golden/real/), read, kekulised
        # and tokenised without RDKit against a vocabulary mined from those same lines; `tok` is that vocabulary
        from ps_vae_amd.data import mol_bpe
        cache = f'/tmp/psvae_{shape}_{seed}.npz'
        if os.path.exists(cache):
            return BPEMolDataset(None, tok, device=device, data=PackedMolecules.load(cache))
        np.random.seed(seed)
        with open(os.path.join(REAL_DIR, f'{shape[:-5]}_2048.smi')) as f:
            data = [mol_bpe.process_step1(line.strip(), tok) for line in f if line.strip()]
        packed = PackedMolecules.from_step1(data)
        try:
            tmp = f'{cache}.{os.getpid()}.tmp.npz'
            packed.save(tmp)
            os.replace(tmp, cache)
        except OSError:
            pass
        return BPEMolDataset(None, tok, device=device, data=packed)
    cache = f'/tmp/psvae_synth_{shape}_{num}_{seed}.npz'
    if os.path.exists(cache):
        packed = PackedMolecules.load(cache)
    else:
        packed = PackedMolecules.from_step1(synthetic_step1(num, seed=seed, shape=shape))
        try:
            tmp = f'{cache}.{os.getpid()}.tmp.npz'      # (another process may be looking for the same cache file)
            packed.save(tmp)
            os.replace(tmp, cache)
        except OSError:
            pass
    return BPEMolDataset(None, tok, device=device, data=packed)


def _cpu_leg(tok, shape, batch, steps, seed):
    """One CPU-oracle leg: `steps` train steps of `batch` molecules (+1 warm-up step); returns
    (collation seconds, model seconds)."""
    from oracle import collate_ref, model_ref
    from ps_vae_amd.data.synthetic import synthetic_step1
    torch.manual_seed(2021)
    model = model_ref.PSVAEModel(model_ref.default_config(tok), tok)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    rng = np.random.default_rng(0)
    mols = synthetic_step1(batch * (steps + 1), seed=seed, shape=shape)
    t_collate = t_model = 0.0
    for s in range(steps + 1):
        chunk = mols[s * batch:(s + 1) * batch]
        t0 = time.perf_counter()
        b, _ = collate_ref.collate(chunk, rng.random(1500 * batch), piece_pad=tok.pad_idx())
        tb = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in b.items()}
        t1 = time.perf_counter()
        model_ref.train_step(model, opt, tb, torch.randn(batch, 56))
        t2 = time.perf_counter()
        if s > 0:            # first step is warm-up
            t_collate += t1 - t0
            t_model += t2 - t1
    return t_collate, t_model


def cpu_baseline(tok, batch, steps, full_batch=0, full_steps=0):
    """The CPU oracle (oracle/: pure-Python collation restatement + plain PyTorch fp32 model,
    clip, Adam) on this machine's host cores, on bounded samples of the same workloads: the
    headline sample at `batch` (default 256), plus batch 32 on ZINC-shaped and QM9-shaped
    molecules (BASELINE.json configs[1] and configs[0], the reference's own CPU case).
    Collation is single-threaded Python, as one DataLoader worker of the reference is."""
    cores = torch.get_num_threads()
    tc, tm = _cpu_leg(tok, 'zinc', batch, steps, 99)
    total = tc + tm
    out = {'value': round(batch * steps / total, 2), 'unit': 'molecules/s', 'cores': cores, 'kind': 'port',
           'sample': f'{steps} train steps of batch {batch} (same synthetic ZINC-shaped generator), '
                     f'collation + forward + backward + clip + Adam, after 1 warm-up step',
           'collate_molecules_per_s_1core': round(batch * steps / tc, 2),
           'model_step_molecules_per_s': round(batch * steps / tm, 2),
           'note': 'kind "port": the oracle is a restatement of the reference, not the reference itself. Its '
                   'collation is a numpy / Python restatement that runs tens of times faster than the '
                   'reference\'s own triple loop over torch tensor elements (about 100 molecules/s per core when '
                   'the reference was profiled), so this baseline is conservative: the reference would be slower.'}
    if full_batch and full_steps:
        tc, tm = _cpu_leg(tok, 'zinc', full_batch, full_steps, 97)
        out[f'b{full_batch}'] = {'value': round(full_batch * full_steps / (tc + tm), 2), 'unit': 'molecules/s', 'cores': cores,
                                 'sample': f'{full_steps} train steps of batch {full_batch} (the batch of the headline '
                                           f'workload), after 1 warm-up step',
                                 'collate_molecules_per_s_1core': round(full_batch * full_steps / tc, 2),
                                 'model_step_molecules_per_s': round(full_batch * full_steps / tm, 2)}
    for name, shape, b, n in (('b32_zinc', 'zinc', 32, 60), ('b32_qm9', 'qm9', 32, 120)):
        tc, tm = _cpu_leg(tok, shape, b, n, 98)
        out[name] = {'value': round(b * n / (tc + tm), 2), 'unit': 'molecules/s', 'cores': cores,
                     'sample': f'{n} train steps of batch {b}, {shape.upper()}-shaped',
                     'collate_molecules_per_s_1core': round(b * n / tc, 2),
                     'model_step_molecules_per_s': round(b * n / tm, 2),
                     'with_4_collation_workers_molecules_per_s': round(b * n / (max(tc / 4, tm)), 2)}
    return out


def pmc_traffic(batch):
    """HBM bytes per launch from the newest committed counter pass (profiles/*_traffic.json:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over this same command, corrected as
    tools/pmc_traffic.py documents) whose batch size equals this run's; returns
    (per-kernel dict or None, description of the source).  The counters come from a separate
    profiled run of the build named in the file, not from this run."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_traffic.json')))
    for path in reversed(files):
        with open(path) as f:
            data = json.load(f)
        meta = data.get('_meta', {})
        if int(meta.get('batch', 4096)) == int(batch):
            return data, {'file': os.path.relpath(path, ROOT), 'build': meta.get('build', 'unknown'),
                          'batch': int(meta.get('batch', 4096)),
                          'note': 'separate rocprofv3 --pmc passes over bench.py at this batch size'}
    return None, None


def small_batch_record(model_factory, tok, shape, B, steps, warmup, dev):
    """A short training run at batch B (BASELINE.json configs[0] / configs[1] shapes): the step recorded
    once in a HIP graph over fixed-size padded buffers and replayed (graph_step.GraphedTrainStep;
    collation of every batch stays outside the recording), beside the same step launched kernel by
    kernel.  The roofline is the whole step's: matrix-product flops of one step (counted on an
    ordinary step of the same model) over the step time."""
    from ps_vae_amd import hip
    from ps_vae_amd.graph_step import GraphedTrainStep
    dataset = make_dataset(64 * B, 7, tok, dev, shape=shape)   # (real molecules: the 2048 of the fixture file)
    rng = np.random.default_rng(3)

    def timed(step_fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step_fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    # the ordinary step
    model, opt = model_factory()

    def eager_step():
        batch = dataset.collate_ids(rng.choice(len(dataset), size=B, replace=False))
        opt.zero_grad()
        model.training_step(batch, 0).backward()
        opt.step()
        model.global_step += 1

    for _ in range(warmup):
        eager_step()
    calls0 = hip.CALLS
    dt_eager = timed(eager_step, steps)
    calls = (hip.CALLS - calls0) // steps
    hip.PROFILE = prof = {}
    eager_step()
    torch.cuda.synchronize()
    hip.PROFILE = None
    flops = sum(it[2] for it in prof.get('gemm_f32_kernel', []))
    gemm_ms = sum(it[0].elapsed_time(it[1]) for it in prof.get('gemm_f32_kernel', []))
    # the recorded step
    model, opt = model_factory()
    graphed = GraphedTrainStep(model, opt, dataset, B)

    def graph_step():
        graphed.step(rng.choice(len(dataset), size=B, replace=False))

    for _ in range(warmup):
        graph_step()
    dt = timed(graph_step, steps)
    tflops = flops / dt / 1e12
    return {'value': round(B / dt, 1), 'unit': 'molecules/s', 'ms_per_step': round(dt * 1e3, 3), 'batch': B, 'shape': shape,
            'steps': steps, 'mode': 'one HIP-graph launch per step (fixed-size padded batch), collation outside the recording',
            'graph_replays': graphed.replays, 'ordinary_fallbacks': graphed.fallbacks, 'padded_shapes': repr(graphed.shapes),
            'kernel_by_kernel': {'value': round(B / dt_eager, 1), 'ms_per_step': round(dt_eager * 1e3, 3),
                                 'abi_calls_per_step': calls},
            'roofline': {'bound': 'mfma', 'scope': 'whole step (all kernels and gaps), matrix-product flops only',
                         'achieved': round(tflops, 2), 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': round(tflops / MFMA_F32_PEAK_TFLOPS, 4), 'gemm_gflop_per_step': round(flops / 1e9, 2),
                         'gemm_kernels_alone_ms': round(gemm_ms, 3),
                         'gemm_kernels_alone_frac': round(flops / (gemm_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4) if gemm_ms else None,
                         'traffic': None}}


def generation_record(model, dev, n_mol=10000, reps=3):
    """BASELINE.json configs[4]: n_mol latents -> piece sampling -> fragment graphs -> link
    prediction over all cross-fragment atom pairs -> greedy assembly (device work of
    VAEPieceDecoder.inference_batch; host packaging of the result lists excluded)."""
    dec = model.decoder
    gen = torch.Generator(device=dev).manual_seed(11)
    z = torch.randn(n_mol, dec.latent_dim, device=dev, generator=gen)
    u = torch.rand(n_mol, 50, device=dev, generator=gen)

    def run():
        ids, lengths = dec.sample_pieces_batch(z, 30, 1.0, uniforms=u)
        g = dec.expand_pieces(ids, lengths)
        links = dec.infer_links_batch(z, g['x'], g['x_pieces'], g['x_pos'], g['node_off'], g['csr_in'], g['block0'], 0.5)
        asm = dec.assemble_links_batch(links, g['x'], g['node_off'], g['csr_in'], g['block0'], g['n_blocks'])
        return g, links, asm

    with torch.no_grad():
        g, links, asm = run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            run()
        torch.cuda.synchronize()
        t_all = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(reps):
            dec.infer_links_batch(z, g['x'], g['x_pieces'], g['x_pos'], g['node_off'], g['csr_in'], g['block0'], 0.5)
        torch.cuda.synchronize()
        t_link = (time.perf_counter() - t0) / reps
        # matrix-product flops of one link prediction, and the time of those kernels alone (event brackets)
        from ps_vae_amd import hip
        hip.PROFILE = prof = {}
        dec.infer_links_batch(z, g['x'], g['x_pieces'], g['x_pos'], g['node_off'], g['csr_in'], g['block0'], 0.5)
        torch.cuda.synchronize()
        hip.PROFILE = None
    flops = sum(it[2] for it in prof.get('gemm_f32_kernel', []))
    gemm_ms = sum(it[0].elapsed_time(it[1]) for it in prof.get('gemm_f32_kernel', []))
    pairs = int(links['conf'].numel())
    tflops = flops / t_link / 1e12
    return {'molecules': n_mol, 'atoms': int(g['x'].numel()), 'candidate_pairs': pairs,
            'value': round(n_mol / t_all, 1), 'unit': 'molecules/s', 'ms_total': round(t_all * 1e3, 2),
            'link_prediction_ms': round(t_link * 1e3, 2), 'link_prediction_pairs_per_s': round(pairs / t_link, 1),
            'roofline': {'bound': 'mfma', 'scope': 'link prediction (all kernels and gaps), matrix-product flops only',
                         'achieved': round(tflops, 2), 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': round(tflops / MFMA_F32_PEAK_TFLOPS, 4), 'gemm_gflop': round(flops / 1e9, 1),
                         'gemm_kernels_alone_ms': round(gemm_ms, 3),
                         'gemm_kernels_alone_frac': round(flops / (gemm_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4) if gemm_ms else None,
                         'traffic': None},
            'note': 'random-init weights, synthetic 300-piece vocabulary; sampling + expansion + link prediction '
                    'over all cross-fragment pairs + greedy assembly'}


def main():
    args = parse()
    if os.environ.get('PSVAE_BENCH_WATCHDOG'):   # dump every thread's stack if the run stalls
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ['PSVAE_BENCH_WATCHDOG']), exit=True)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP path has no CPU fallback)')
    dev_index = local_rank if args.device < 0 else args.device
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    if world > 1:
        if args.backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    import ps_vae_amd  # noqa: F401
    from ps_vae_amd import hip
    from ps_vae_amd.pl_models import PSVAEModel
    from ps_vae_amd.utils.nn_utils import build_config, default_args
    from ps_vae_amd.utils.tokenizer import Tokenizer

    tok = Tokenizer()
    cfg_args = default_args()
    torch.manual_seed(2021)                       # identical initial weights on every rank
    model = PSVAEModel(build_config(cfg_args, tok), tok).to(dev)
    (opt,), _ = model.configure_optimizers()      # optim.FlatAdam: all-reduce + clip + Adam on flat buffers
    opt.max_norm = cfg_args.grad_clip
    B = args.batch
    dataset = make_dataset(args.dataset or 4 * B, 2021 + rank, tok, dev)
    rng = np.random.default_rng(2021 + rank)
    torch.manual_seed(2021 + rank)                # per-rank streams for eps / negative sampling

    def step():
        ids = rng.choice(len(dataset), size=B, replace=False)
        batch = dataset.collate_ids(ids)
        opt.zero_grad()
        model.training_step(batch, 0).backward()
        opt.step()
        model.global_step += 1
        return batch

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # a full collection over the interpreter's ~1e6 module objects costs ~100 ms; take it after the first
    # warm-up step and freeze the survivors, so that it can land neither inside the timed region nor right
    # in front of it (the GPU would sit idle for that long and start the timed steps from its idle clocks)
    for w in range(args.warmup):
        step()
        if w == 0:
            gc.collect()
            gc.freeze()
    if args.warmup == 0:
        gc.collect()
        gc.freeze()
    # Kernel events are recorded on a few of the timed steps only (they cost ~0.6 ms of host
    # time per step).  Steps 4, 12, 20, ... run with everything on ONE stream and fill `prof`:
    # their brackets time each kernel alone.  Steps 8, 24, ... keep the two streams and fill
    # `prof_shared`: there kernels of both streams share the chip and every bracket stretches.
    # (Not step 0: the first step behind the idle fence runs its products ~10 % slower than the steps
    # of a running loop - 84 against 94 TFLOP/s on the conv shapes - and is not what a training run sees.)
    prof, prof_shared = {}, {}
    two_streams = model.two_streams
    fence()
    t0 = time.perf_counter()
    stats = {'V': 0, 'E': 0, 'Ein': 0, 'P': 0}
    trace = []
    # a step ends with the optimiser kernel on the main stream, behind everything else of the step:
    # an event there marks the step's end on the GPU's own clock
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    bracketed = []
    marks[0].record()
    for i in range(args.steps):
        alone = i % 8 == min(4, args.steps - 1)      # (a run of fewer than five steps samples its last one)
        hip.PROFILE = prof if alone else (prof_shared if i % 16 == 8 else None)
        bracketed.append(hip.PROFILE is not None)
        model.two_streams = two_streams and not alone
        b = step()
        marks[i + 1].record()
        trace.append(time.perf_counter())
        c = b.compact
        stats['V'] += c.num_nodes
        stats['E'] += c.csr.col.numel()
        stats['Ein'] += c.csr_in.col.numel()
        stats['P'] += len(c.pairs)
    fence()
    elapsed = time.perf_counter() - t0
    if os.environ.get('PSVAE_BENCH_TRACE'):
        print('host ms per step (no sync):', [round((b - a) * 1e3, 1) for a, b in zip([t0] + trace[:-1], trace)],
              'final fence ms', round((t0 + elapsed - trace[-1]) * 1e3, 1), file=sys.stderr)
    hip.PROFILE = None
    rccl_ranks = None
    if world > 1 and args.backend == 'nccl':
        # how many ranks a collective on the GPUs actually joins: one all-reduce of a one per rank
        ones = torch.ones(1, dtype=torch.float32, device=dev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        rccl_ranks = int(ones.item())
    final_loss = float(model.logged['reconstruct loss'].detach())
    model.two_streams = two_streams
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    plain = [t for t, br in zip(step_ms, bracketed) if not br]
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        out = {
            'metric': 'molecules/sec per PS-VAE train step', 'value': round(value, 1), 'unit': 'molecules/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
            'data': 'synthetic ZINC-shaped molecules (~23 atoms, ~24 bonds), random-init weights',
            'config': {'workload': f'BASELINE.json configs[2]: synthetic ZINC-shaped graphs, batch {B} per GPU, '
                                   'full train step (GPU collation + encoder + latent + link predictor + piece LM '
                                   '+ backward + all-reduce + clip + Adam)',
                       'batch_per_gpu': B, 'global_batch': B * world, 'parallelism': f'dp{world}',
                       'atoms_per_step': stats['V'] // args.steps, 'pairs_per_step': stats['P'] // args.steps},
            'rec_loss_last_step': round(final_loss, 5),
            # GPU-clock duration of the timed steps that carry no kernel event brackets (the brackets cost
            # host time and some of those steps run one stream); `value` and `ms_per_step` cover ALL steps
            'ms_per_step_without_event_brackets': round(sum(plain) / len(plain), 3) if plain else None,
            'steps_with_event_brackets': sum(bracketed),
        }
        if world > 1:
            out['rccl_ranks'] = rccl_ranks          # ranks that joined an all-reduce over RCCL (None: another backend)
            out['collective'] = f'one flat all-reduce of {opt.sync.flat.numel()} f32 per step over {dist.get_backend()}'
        traffic, traffic_source = pmc_traffic(B)
        out.update(hip.roofline_report(prof, HBM_PEAK_GBS, MFMA_F32_PEAK_TFLOPS, traffic))
        out['traffic_source'] = traffic_source
        shared = hip.roofline_report(prof_shared, HBM_PEAK_GBS, MFMA_F32_PEAK_TFLOPS, traffic)
        out['roofline_two_streams'] = {k: shared[k] for k in ('roofline_mfma', 'roofline_hbm', 'roofline_hbm_bwd')
                                       if k in shared}
        out['roofline_note'] = ('roofline*: event brackets on timed steps 4, 12, 20, ..., which run on one '
                                'stream so that each bracket times its kernel alone; roofline_two_streams: the '
                                'same brackets on timed steps 8, 24, ..., where kernels of the two streams share '
                                'the chip; all other timed steps run two streams without events')
        if args.gemm_shapes:
            print('M N K A-contig B-contig masked rowsum launches ms TFLOP/s', file=sys.stderr)
            for row in hip.gemm_shape_table(prof):
                print(' '.join(str(v) for v in row), file=sys.stderr)
        if world == 1 and not args.no_sub_records:
            # the other single-GPU configurations of BASELINE.json on the same build (not the headline)
            def factory():
                torch.manual_seed(2021)
                m = PSVAEModel(build_config(cfg_args, tok), tok).to(dev)
                (o,), _ = m.configure_optimizers()
                o.max_norm = cfg_args.grad_clip
                return m, o
            del dataset
            out['b32'] = small_batch_record(factory, tok, 'zinc', 32, 200, 30, dev)
            out['b32_qm9'] = small_batch_record(factory, tok, 'qm9', 32, 200, 30, dev)
            for shape in ('zinc', 'qm9'):          # configs[1] / configs[0] on real molecules
                rtok = real_tokenizer(shape)

                def real_factory():
                    torch.manual_seed(2021)
                    m = PSVAEModel(build_config(cfg_args, rtok), rtok).to(dev)
                    (o,), _ = m.configure_optimizers()
                    o.max_norm = cfg_args.grad_clip
                    return m, o
                rec = small_batch_record(real_factory, rtok, shape + '_real', 32, 200, 30, dev)
                rec['data'] = (f'real molecules: 2048 lines of the reference\'s data/{"qm9/train" if shape == "qm9" else "zinc250k/valid"}'
                               '.txt, preprocessed without RDKit (own kekulisation, 300-piece vocabulary mined from the '
                               'same lines, property scores zero)')
                out[f'b32_{shape}_real'] = rec
            out['infer_10k'] = generation_record(model, dev)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(tok, args.cpu_batch, args.cpu_steps, B, args.cpu_full_steps)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
