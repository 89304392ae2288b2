"""Device / process-group bootstrap (ref: framework/device.py:21-166).

Same env discovery (torchrun / SLURM / MPI / PMI), same ``DeviceEnv`` fields and collectives.
Differences, both deliberate: (1) a CPU device type exists (the enum already lists it,
device.py:48-53) so host logic and gloo tests run without a GPU -- compute still requires HIP;
(2) backend "nccl" binds RCCL on ROCm; on a CPU env the backend falls back to gloo.
"""
import os
from dataclasses import InitVar, dataclass, field
from enum import Enum
from typing import Optional

import torch
import torch.distributed as dist


def is_distributed_env():
    if 'WORLD_SIZE' in os.environ:
        return int(os.environ['WORLD_SIZE']) > 1
    if 'SLURM_NTASKS' in os.environ:
        return int(os.environ['SLURM_NTASKS']) > 1
    return False


def launched_by_torchrun():
    """torchrun / torch.distributed.run exports all of these, also for a single rank: such a process takes the distributed
    branch (RCCL communicator, bucketed reducer) even when WORLD_SIZE == 1, so `bench.py --gpus 1` under the launcher
    walks exactly the code `--gpus 8` does."""
    return all(v in os.environ for v in ('RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'))


def world_info_from_env():
    local_rank = 0
    for v in ('LOCAL_RANK', 'MPI_LOCALRANKID', 'SLURM_LOCALID', 'OMPI_COMM_WORLD_LOCAL_RANK'):
        if v in os.environ:
            local_rank = int(os.environ[v])
            break
    global_rank = 0
    for v in ('RANK', 'PMI_RANK', 'SLURM_PROCID', 'OMPI_COMM_WORLD_RANK'):
        if v in os.environ:
            global_rank = int(os.environ[v])
            break
    world_size = 1
    for v in ('WORLD_SIZE', 'PMI_SIZE', 'SLURM_NTASKS', 'OMPI_COMM_WORLD_SIZE'):
        if v in os.environ:
            world_size = int(os.environ[v])
            break
    return local_rank, global_rank, world_size


class DeviceEnvType(Enum):
    CPU = 'cpu'
    CUDA = 'cuda'
    XLA = 'xla'


@dataclass
class DeviceEnv:
    init_device_type: InitVar[Optional[str]] = None
    init_device_index: InitVar[Optional[int]] = None
    init_dist_backend: InitVar[str] = 'nccl'
    init_dist_url: InitVar[str] = 'env://'

    device: torch.device = field(init=False)
    world_size: Optional[int] = None
    local_rank: Optional[int] = None
    global_rank: Optional[int] = None
    distributed: bool = False   # a process group exists (world_size > 1, or a single rank started by torchrun)

    def is_global_primary(self):
        return self.global_rank == 0

    def is_local_primary(self):
        return self.local_rank == 0

    def is_primary(self, local=False):
        return self.is_local_primary() if local else self.is_global_primary()

    def __post_init__(self, init_device_type, init_device_index, init_dist_backend, init_dist_url):
        if init_device_type is None:
            init_device_type = 'cuda' if torch.cuda.device_count() else 'cpu'
        use_cuda = init_device_type == 'cuda'
        if use_cuda:
            assert torch.cuda.device_count(), 'no HIP device visible'
        init_local_rank, init_global_rank, init_world_size = world_info_from_env()
        if init_world_size > 1 or launched_by_torchrun():
            assert init_device_index is None
            self.local_rank = int(init_local_rank)
            backend = init_dist_backend if use_cuda else 'gloo'
            # validation aid for 1-GPU boxes (scripts/check_dp2_shared_gpu.py): every rank on device 0, collectives over
            # gloo (RCCL refuses two ranks on one device). Never set in production launches.
            shared_gpu = use_cuda and os.environ.get('CRL_DEBUG_SHARED_GPU', '0') == '1'
            if shared_gpu:
                backend = 'gloo'
            device_index = 0 if shared_gpu else self.local_rank
            if use_cuda:
                torch.cuda.set_device(device_index)  # before the RCCL communicator is created
            if not dist.is_initialized():
                if 'SLURM_PROCID' in os.environ:
                    dist.init_process_group(backend=backend, init_method=init_dist_url, world_size=init_world_size,
                                            rank=init_global_rank)
                else:
                    dist.init_process_group(backend=backend, init_method=init_dist_url)
            self.world_size = dist.get_world_size()
            self.global_rank = dist.get_rank()
            self.distributed = True
            self.device = torch.device('cuda:%d' % device_index) if use_cuda else torch.device('cpu')
        else:
            if use_cuda:
                self.device = torch.device('cuda' if init_device_index is None else f'cuda:{init_device_index}')
            else:
                self.device = torch.device('cpu')
            self.local_rank = 0
            self.world_size = 1
            self.global_rank = 0

    def broadcast_object(self, obj, src=0):
        if self.world_size == 1:
            return obj
        objects = [obj] if self.global_rank == src else [None]
        dist.broadcast_object_list(objects, src=src)
        return objects[0]

    def all_gather_object(self, obj, dst=0):
        if self.world_size == 1:
            return [obj]
        objects = [None for _ in range(self.world_size)]
        dist.all_gather_object(objects, obj)
        return objects
