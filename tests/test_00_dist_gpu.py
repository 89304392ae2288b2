"""Data-parallel path on the GPU box (`-m gpu`): the RCCL branch of DeviceEnv / BucketedGradReducer on HIP tensors, and the
two-rank equivalence DP2 == grad-accum 2.  The box has ONE MI355X, so
  * the RCCL test runs a single rank with a real `nccl` process group (tests/_rccl_child.py, a fresh interpreter started
    with the torchrun environment before anything touches the GPU) -- ordering compute -> collective -> optimiser, no_sync,
    parameter broadcast, and a bit-exact train_step through the bucketed asynchronous all-reduce;
  * the two-rank test puts both ranks on the one device with collectives over gloo (RCCL refuses two ranks per device):
    scripts/check_dp2_shared_gpu.py, the whole N > 1 control flow against one process with grad_accum_steps = 2.
ref: task/task_cruller_pretrain.py:181-189,280-283 (DistributedDataParallel + no_sync), framework/device.py:116-135."""
import os
import socket
import subprocess
import sys

import pytest
import torch          # device_count() below does not initialise the GPU: the child processes are the first to touch it

pytestmark = pytest.mark.gpu
N_GPUS = torch.cuda.device_count()
needs_two_gpus = pytest.mark.skipif(N_GPUS < 2, reason=f'needs two MI355X for a real N > 1 RCCL run (this box has {N_GPUS}); '
                                                         'RCCL refuses two ranks on one device')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _env(**kw):
    e = dict(os.environ)
    e.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    e.update({k: str(v) for k, v in kw.items()})
    return e


def _run(cmd, timeout, env):
    """run a child launcher; these take 5-25 s.  A launch that has not finished after `timeout` seconds is killed and started ONCE more
    (round 4: one run of the suite lost 15 minutes to a two-rank torchrun child that never got past its gloo rendezvous, on a box where
    the same test takes 7 s; the retry keeps a stuck rendezvous from eating the whole suite's time budget -- a second hang fails the test).
    A child that exits non-zero is also started once more (round 6, see below)."""
    import signal
    cmd = list(cmd)
    for attempt in (1, 2):
        if attempt == 2 and '--master-port' in cmd:
            cmd[cmd.index('--master-port') + 1] = str(_free_port())
        if attempt == 2 and 'MASTER_PORT' in env:
            env = dict(env, MASTER_PORT=str(_free_port()))
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env, start_new_session=True)
        try:
            out, err = p.communicate(timeout=timeout)
            if p.returncode != 0 and attempt == 1:
                # round 6: one run of the suite lost its FIRST test in 8 s to a child that died during start-up on a fresh box (the same tree passed on
                # the boxes before and after it).  A launcher that fails is started once more, like one that hangs; a real defect fails twice, and the
                # first failure stays visible in the captured output
                print(f'[dist test] attempt 1: {" ".join(map(str, cmd[-4:]))} exited with {p.returncode}; stdout tail: {out[-500:]} stderr tail: {err[-1500:]}')
                continue
            return subprocess.CompletedProcess(cmd, p.returncode, out, err)
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, signal.SIGKILL)          # the launcher AND its workers (its own process group: started with a new session)
            out, err = p.communicate()
            print(f'[dist test] attempt {attempt}: {" ".join(map(str, cmd[-4:]))} timed out after {timeout} s; stderr tail: {err[-500:]}')
            if attempt == 2:
                raise


def test_rccl_single_rank_reducer_and_train_step():
    r = _run([sys.executable, os.path.join(ROOT, 'tests', '_rccl_child.py')], 300,
             _env(RANK=0, LOCAL_RANK=0, WORLD_SIZE=1, MASTER_ADDR='127.0.0.1', MASTER_PORT=_free_port()))
    assert r.returncode == 0 and 'RCCL_CHILD_OK' in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    print(r.stdout[-300:])


def test_bench_under_torchrun_single_rank_takes_rccl_path():
    """`torchrun --nproc-per-node 1 bench.py --gpus 1` = the driver's N > 1 command line at N = 1: process group, reducer
    and the rank-0 JSON line all come from the code `--gpus 8` runs"""
    import json
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1',
           '--model', 'cruller_base_960x640', '--batch', '2', '--no-cpu-baseline', '--no-roofline']
    r = _run(cmd, 400, _env())
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert r.returncode == 0 and lines, (r.stdout[-2000:], r.stderr[-4000:])
    d = json.loads(lines[-1])
    assert d['n_gpus'] == 1 and d['value'] > 0 and d['config']['parallelism'] == 'dp1' and d.get('collectives') == 'rccl'
    # the diagnosis of a multi-GPU run rides on the same line: bucket geometry, CUs reserved for RCCL, exposed
    # communication per optimiser step with min / max over the ranks, per-rank step time
    c = d['comm']
    assert c['buckets'] >= 1 and c['bucket_bytes'] == 64 << 20 and c['reductions'] == d['steps'] and c['reserved_cus'] == 0     # one rank: nothing to reserve
    assert c['comm_exposed_ms_max'] >= c['comm_exposed_ms'] >= 0.0 and c['comm_exposed_ms_rank_max'] >= c['comm_exposed_ms_rank_min'] >= 0.0
    assert c['step_ms_rank_max'] >= c['step_ms_rank_min'] > 0.0 and abs(c['step_ms_rank_max'] - d['ms_per_step']) < 0.02


def test_dp2_on_shared_gpu_equals_grad_accum_2(tmp_path):
    out = str(tmp_path / 'dp2.pt')
    script = os.path.join(ROOT, 'scripts', 'check_dp2_shared_gpu.py')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), script, '--out', out]
    r = _run(cmd, 300, _env(CRL_DEBUG_SHARED_GPU=1))
    assert r.returncode == 0 and os.path.exists(out), (r.stdout[-2000:], r.stderr[-4000:])
    r = _run([sys.executable, script, '--reference', out], 300, _env())
    assert r.returncode == 0 and 'DP2 == ACCUM2: OK' in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@needs_two_gpus
def test_rccl_two_ranks_equals_accum2(tmp_path):
    """cfg-4 in miniature, the day a multi-GPU box runs this suite: two ranks on two devices, backend `nccl` (= RCCL over xGMI), through
    the bucketed asynchronous gradient reducer (fresh torchrun processes: nothing has touched the GPUs before the process group exists),
    compared with ONE process running grad_accum_steps = 2 over the same four batches -- parameters, AdamW m / v, losses.
    ref: task/task_cruller_pretrain.py:181-189,280-283; framework/device.py:116-135."""
    out = str(tmp_path / 'dp2_rccl.pt')
    script = os.path.join(ROOT, 'scripts', 'check_dp2_shared_gpu.py')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), script, '--out', out, '--expect-backend', 'nccl']
    env = _env()
    env.pop('CRL_DEBUG_SHARED_GPU', None)
    r = _run(cmd, 300, env)
    assert r.returncode == 0 and os.path.exists(out), (r.stdout[-2000:], r.stderr[-4000:])
    r = _run([sys.executable, script, '--reference', out], 300, env)
    assert r.returncode == 0 and 'DP2 == ACCUM2: OK' in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@needs_two_gpus
def test_bench_two_ranks_rccl():
    """the driver's N = 2 command line on a small config: both ranks finish, rank 0 prints one line, value counts both ranks' docs"""
    import json
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
           '--model', 'cruller_base_960x640', '--batch', '2', '--no-cpu-baseline', '--no-roofline']
    r = _run(cmd, 400, _env())
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-4000:])
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['global_batch'] == 4 and d['collectives'] == 'rccl' and d['value'] > 0
