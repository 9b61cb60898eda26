"""Work distribution (``zuds/mpi.py``): the same ``np.array_split`` sharding of
a job list, over ``torch.distributed`` (or the launcher's RANK / WORLD_SIZE
environment) instead of mpi4py's ``comm.scatter``."""
import os

import numpy as np

from .constants import NTHREADS_PER_NODE

__all__ = ['get_nthreads', 'get_my_share_of_work', 'has_mpi']


def default_reader(f):
    return np.atleast_1d(np.genfromtxt(f, dtype=None, encoding='ascii'))


def _rank_size():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass
    if 'RANK' in os.environ and 'WORLD_SIZE' in os.environ:
        return int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    return 0, 1


def get_nthreads():
    slurm_count = os.getenv('SLURM_CPUS_PER_TASK')
    return NTHREADS_PER_NODE if slurm_count is None else slurm_count


def has_mpi():
    return _rank_size()[1] > 1


def get_my_share_of_work(fname, reader=default_reader):
    """Rank's slice of the job list: split by Slurm array task, then by rank
    (``zuds/mpi.py:36-64``).  Every rank reads the (kB-sized) list itself, which
    yields the same partition as rank 0 scattering it."""
    rank, size = _rank_size()
    files = reader(fname)
    if os.getenv('SLURM_ARRAY_JOB_ID') is not None:
        job_array_index = int(os.getenv('SLURM_ARRAY_TASK_ID'))
        job_array_ntasks = int(os.getenv('SLURM_ARRAY_TASK_MAX')) + 1
        files = np.array_split(files, job_array_ntasks)[job_array_index]
    if size == 1:
        return files
    return np.array_split(files, size)[rank]
