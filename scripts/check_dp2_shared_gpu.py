"""Data-parallel equivalence check on a ONE-GPU box (the 8-GPU node is the driver's):
  CRL_DEBUG_SHARED_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
      --master-port 29533 scripts/check_dp2_shared_gpu.py --out /tmp/dp2.pt
  python scripts/check_dp2_shared_gpu.py --reference /tmp/dp2.pt
The first command runs TaskCrullerPretrain on two ranks (both on cuda:0, collectives over gloo) through the bucketed
asynchronous gradient reducer (without CRL_DEBUG_SHARED_GPU on a box with two devices the same command runs the real thing: one device per
rank, backend nccl = RCCL -- tests/test_00_dist_gpu.py::test_rccl_two_ranks_equals_accum2); the second runs ONE process with grad_accum_steps=2 over the same four batches -- the
same average of two per-batch mean-loss gradients -- and compares parameters, AdamW state and losses."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


VIT = dict(patch=8, dim=128, depth=2, heads=2, mlp_ratio=4, ln_eps=1e-6, pre_norm=False, mean=(0.5,) * 3, std=(0.5,) * 3)
BART = dict(d_model=128, heads=2, ffn=256, ln_eps=1e-5, vocab=509, dropout=0.0)


def make_task(accum):
    from pixparse_amd.framework import DeviceEnv, OptimizationCfg
    from pixparse_amd.models import ImageEncoderCfg, ModelCfg, TextDecoderCfg
    from pixparse_amd.models.archs import register_arch
    from pixparse_amd.task import TaskCrullerPretrain, TaskCrullerPretrainCfg
    register_arch('vit', 'vit_dp_check', VIT)
    register_arch('bart', 'bart_dp_check', BART)
    model = ModelCfg(image_encoder=ImageEncoderCfg(name='vit_dp_check', image_fmt='RGB', image_size=(37, 50), pretrained=False),
                     text_decoder=TextDecoderCfg(name='bart_dp_check', pretrained=False, num_decoder_layers=2, max_length=24))
    cfg = TaskCrullerPretrainCfg(num_intervals=4, num_warmup_intervals=1, eval_frequency=1000, dtype='bfloat16',
                                 opt=OptimizationCfg(learning_rate=1e-3, betas=(0.9, 0.98), clip_grad_value=1.0, clip_grad_mode='norm',
                                                     grad_accum_steps=accum),
                                 model=model)
    torch.manual_seed(5)
    env = DeviceEnv()
    t = TaskCrullerPretrain(cfg, env)
    t.train_setup(num_batches_per_interval=2 * accum)
    t.train_interval_start()
    return t, env


def samples():
    from pixparse_amd.data import synthetic_batch
    return [synthetic_batch(2, 3, (37, 50), 24, 50267, seed=40 + i, ragged=True) for i in range(4)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out')
    ap.add_argument('--reference')
    ap.add_argument('--expect-backend', default=None, help="fail unless the process group runs on this backend ('nccl' = RCCL: two real devices)")
    a = ap.parse_args()
    ss = samples()
    if a.reference:
        t, env = make_task(accum=2)
        assert env.world_size == 1
        losses = []
        for s in ss:
            t.train_step(s)
            losses.append(float(t.last_loss))
        got = torch.load(a.reference)
        ar = t.model.arena
        for name, mine in (('p', ar.p), ('m', ar.m), ('v', ar.v)):
            ref = got[name].to(mine.device)
            err = float((mine - ref).abs().max()); scale = float(ref.abs().max())
            print(f'{name}: max |dp2 - accum2| = {err:.3e} (max |value| {scale:.3e})')
            assert err <= 2e-6 * max(scale, 1.0) + 1e-9, name
        # per-step loss of the DP run = mean of the two ranks' losses = sum of the two accumulation micro-losses
        dp_losses = got['losses']
        acc_losses = [losses[0] + losses[1], losses[2] + losses[3]]
        print('losses dp2', dp_losses, 'accum2', acc_losses)
        assert all(abs(x - y) < 1e-5 * abs(y) for x, y in zip(dp_losses, acc_losses))
        print('DP2 == ACCUM2: OK')
        return
    import torch.distributed as dist
    t, env = make_task(accum=1)
    assert env.world_size == 2, 'launch with torchrun --nproc-per-node 2'
    if a.expect_backend:
        assert dist.get_backend() == a.expect_backend, f'process group runs on {dist.get_backend()!r}, expected {a.expect_backend!r}'
        assert torch.cuda.current_device() == env.local_rank or env.device.index == env.local_rank, 'one device per rank'
    losses = []
    for k in range(2):
        t.train_step(ss[2 * k + env.global_rank])
        l = torch.tensor([float(t.last_loss)], dtype=torch.float64, device=env.device if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(l)
        losses.append(float(l) / 2)
    ar = t.model.arena
    other = [torch.empty_like(ar.p) for _ in range(2)]
    dist.all_gather(other, ar.p)
    assert torch.equal(other[0], other[1]), 'ranks diverged'
    if env.global_rank == 0:
        torch.save({'p': ar.p.cpu(), 'm': ar.m.cpu(), 'v': ar.v.cpu(), 'losses': losses}, a.out)
        print('dp2 run done: ranks identical, losses', losses)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
