"""Child process of tests/test_00_dist_gpu.py: ONE rank with a real RCCL communicator (backend "nccl") on cuda:0.

Started as a fresh interpreter with the torchrun environment (RANK / WORLD_SIZE=1 / MASTER_*), so the process group is
created before anything else touches the GPU. Exercises the production data-parallel code on HIP tensors:
  1. DeviceEnv takes the distributed branch for a single torchrun rank (framework/device.py);
  2. BucketedGradReducer begin / on_ready / finish with compute kernels still enqueued on the compute stream when the
     collectives are issued: the all-reduce must see the FINAL gradient (compute -> comm ordering) and the kernels that
     follow finish() must see the reduced one (comm -> compute ordering). A one-rank SUM is the identity and would hide
     both, so the reducer is given a pre-multiplied sum (x2) for this check;
  3. broadcast_params on the parameter arena;
  4. TaskCrullerPretrain.train_step through the RCCL path (bucketed async all-reduce inside backward, no_sync
     micro-steps) against the same task without a process group: a one-rank SUM must not change a single bit.
Prints RCCL_CHILD_OK on success.  ref: task/task_cruller_pretrain.py:181-189,280-283; framework/device.py:116-135.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def check_reducer(env):
    from pixparse_amd.framework.reducer import BucketedGradReducer
    from pixparse_amd.layers.arena import ParamArena
    dev = env.device
    arena = ParamArena()
    n_entries, numel = 12, 1 << 20
    for i in range(n_entries):
        arena.add(f'e{i}', (numel,))
    arena.materialize(dev)
    arena.alloc_training_state()
    try:
        op = dist._make_nccl_premul_sum(2.0)
    except Exception as e:  # noqa: BLE001
        print('premul sum unavailable:', repr(e))
        op = None
    red = BucketedGradReducer(arena, env.world_size, bucket_bytes=3 * numel * 4, active=True,
                              op=op if op is not None else dist.ReduceOp.SUM)
    assert len(red.buckets) == 4 and red.buckets[0][1] == arena.total
    a = torch.randn(4096, 4096, device=dev)
    final = torch.arange(arena.total, device=dev, dtype=torch.float32) % 977.0 + 1.0
    for rep in range(3):
        arena.g.fill_(-5.0)                          # stale values: what a too-early collective would reduce
        torch.cuda.synchronize()
        red.begin()
        for i in reversed(range(n_entries)):         # backward order: the arena completes from its end
            x = a
            for _ in range(6):                       # ~1 ms of compute in front of every gradient write
                x = x @ a
                x = x / x.abs().max()
            e = arena.entries[f'e{i}']
            arena.g[e.offset:e.offset + e.numel].copy_(final[e.offset:e.offset + e.numel] + 0.0 * x[0, 0])
            red.on_ready(f'e{i}')
        red.finish()
        seen = arena.g.clone()                       # enqueued on the compute stream right behind finish()
        torch.cuda.synchronize()
        want = final * (2.0 if op is not None else 1.0)
        assert torch.equal(seen, want), f'rep {rep}: {int((seen != want).sum())} elements differ (ordering violated)'
    # no_sync micro-step: nothing may be reduced
    red.enabled = False
    arena.g.copy_(final)
    red.begin(); red.on_ready('e0'); red.finish()
    torch.cuda.synchronize()
    assert torch.equal(arena.g, final) and red.grad_divisor() == 1.0
    red.enabled = True
    arena.p.copy_(final)
    red.broadcast_params(0)
    torch.cuda.synchronize()
    assert torch.equal(arena.p, final)
    return op is not None


def check_task(env):
    from pixparse_amd.data import synthetic_batch
    from pixparse_amd.framework import OptimizationCfg
    from pixparse_amd.models import ImageEncoderCfg, ModelCfg, TextDecoderCfg
    from pixparse_amd.models.archs import register_arch
    from pixparse_amd.task import TaskCrullerPretrain, TaskCrullerPretrainCfg
    register_arch('vit', 'vit_rccl', dict(patch=8, dim=128, depth=2, heads=2, mlp_ratio=4, ln_eps=1e-6, pre_norm=False, mean=(0.5,) * 3, std=(0.5,) * 3))
    register_arch('bart', 'bart_rccl', dict(d_model=128, heads=2, ffn=256, ln_eps=1e-5, vocab=509, dropout=0.0))
    model = ModelCfg(image_encoder=ImageEncoderCfg(name='vit_rccl', image_fmt='RGB', image_size=(37, 50), pretrained=False),
                     text_decoder=TextDecoderCfg(name='bart_rccl', pretrained=False, num_decoder_layers=2, max_length=24))

    class LocalEnv:                                   # the same device without a process group
        device, world_size, local_rank, global_rank, distributed = env.device, 1, 0, 0, False

    def run(e):
        cfg = TaskCrullerPretrainCfg(num_intervals=2, num_warmup_intervals=1, eval_frequency=1000, dtype='bfloat16',
                                     opt=OptimizationCfg(learning_rate=1e-3, betas=(0.9, 0.98), clip_grad_value=1.0, clip_grad_mode='norm',
                                                         grad_accum_steps=2),
                                     model=model)
        torch.manual_seed(5)
        t = TaskCrullerPretrain(cfg, e)
        t.train_setup(num_batches_per_interval=4)
        t.train_interval_start()
        assert t.reducer.active == e.distributed and t.has_no_sync == e.distributed
        if e.distributed:                             # 64 MiB buckets would put this toy arena into one: use several
            from pixparse_amd.framework.reducer import BucketedGradReducer
            t.reducer = BucketedGradReducer(t.model.arena, 1, bucket_bytes=1 << 20, active=True)
            assert len(t.reducer.buckets) > 8
        out = []
        for i in range(4):
            t.train_step(synthetic_batch(2, 3, (37, 50), 24, 50267, seed=40 + i, ragged=True))
            out.append((float(t.last_loss), float(t.optimizer.grad_norm())))
        torch.cuda.synchronize()
        return out, t.model.arena.p.clone()
    la, pa = run(env)
    lb, pb = run(LocalEnv)
    assert la == lb, (la, lb)
    if not torch.equal(pa, pb):
        d = (pa - pb).abs()
        idx = torch.nonzero(d > 0).flatten()
        raise AssertionError(f'{idx.numel()} of {pa.numel()} parameters differ, max |diff| {float(d.max()):.3e}, first at {idx[:8].tolist()}')


def main():
    from pixparse_amd.framework import DeviceEnv
    env = DeviceEnv()
    assert env.distributed and env.world_size == 1 and dist.is_initialized() and dist.get_backend() == 'nccl', \
        (env, dist.is_initialized())
    premul = check_reducer(env)
    check_task(env)
    dist.barrier()
    dist.destroy_process_group()
    print('RCCL_CHILD_OK premul_sum=%d' % int(premul))


if __name__ == '__main__':
    main()
