"""Does counting the GPUs initialise the HIP runtime in this process?  (bench.py's launcher / supervisor parents must never hold a GPU
context: they kill and restart their children.)  Prints the /dev/kfd and /dev/dri descriptors the process holds before and after
torch.cuda.device_count(), torch.cuda.is_initialized(), and what bench._count_gpus() (sysfs, no runtime) says."""
import os
import sys


def gpu_fds():
    out = []
    for f in os.listdir('/proc/self/fd'):
        try:
            t = os.readlink(f'/proc/self/fd/{f}')
        except OSError:
            continue
        if 'kfd' in t or '/dri/' in t:
            out.append(t)
    return sorted(out)


import torch  # noqa: E402
print('before:', gpu_fds())
n = torch.cuda.device_count()
print('torch.cuda.device_count() =', n, '| torch.cuda.is_initialized() =', torch.cuda.is_initialized())
print('after :', gpu_fds())
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
print('bench._count_gpus() (sysfs) =', bench._count_gpus(), '| host cpus', os.cpu_count())
print('after sysfs count:', gpu_fds())
