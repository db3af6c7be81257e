"""Host-side hardware probes that never touch the HIP runtime.

A launcher that fork+execs one rank per GPU (bench.py --gpus N, the job server) must not have initialised
the GPU itself, so the GPU count comes from the kernel driver's topology files and the visibility
environment variables, not from ``torch.cuda`` (whose amdsmi route falls back to hipGetDeviceCount).
"""
import glob
import os

KFD_NODES = '/sys/class/kfd/kfd/topology/nodes'
DRI = '/dev/dri'


def _visible_list(env):
    """Entries of a *_VISIBLE_DEVICES variable, or None when it is unset.  An empty value hides every device."""
    for name in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = env.get(name)
        if v is not None:
            return [s for s in (t.strip() for t in v.split(',')) if s]
    return None


def kfd_gpu_nodes(root=KFD_NODES):
    """KFD topology nodes that are GPUs (simd_count > 0; CPU nodes report 0).  [] when the driver is absent."""
    nodes = []
    for prop in sorted(glob.glob(os.path.join(root, '*', 'properties')),
                       key=lambda p: int(os.path.basename(os.path.dirname(p)))):
        try:
            with open(prop) as f:
                fields = dict(line.split(None, 1) for line in f if len(line.split(None, 1)) == 2)
            if int(fields.get('simd_count', '0').strip()) > 0:
                nodes.append(int(os.path.basename(os.path.dirname(prop))))
        except (OSError, ValueError):
            continue
    return nodes


def render_nodes(dri=DRI):
    """DRM render nodes this process may open (a container is handed only the nodes of its own GPUs; the KFD
    topology in sysfs still lists the whole host).  None when /dev/dri does not exist."""
    if not os.path.isdir(dri):
        return None
    return [p for p in sorted(glob.glob(os.path.join(dri, 'renderD*'))) if os.access(p, os.R_OK | os.W_OK)]


def count_gpus(env=None, root=KFD_NODES, dri=DRI):
    """Number of GPUs a child process started with `env` would see, without a single HIP / torch.cuda call:
    the accessible render nodes, clipped by the KFD topology's GPU nodes and by a *_VISIBLE_DEVICES list when one is
    set.  Returns None when no source exists (the caller then lets the ranks fail on set_device and propagates
    their exit code): the count only ever refuses a launch that could not have worked."""
    env = os.environ if env is None else env
    bounds = []
    rn = render_nodes(dri)
    if rn is not None:
        bounds.append(len(rn))
    if os.path.isdir(root):
        bounds.append(len(kfd_gpu_nodes(root)))
    vis = _visible_list(env)
    if vis is not None:
        bounds.append(len(vis))
    return min(bounds) if bounds else None
