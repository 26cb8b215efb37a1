"""One process per GPU, started from a parent that never touches a GPU.

The reference drives N devices from one Python thread (``gluon.utils.split_and_load`` +
``Trainer(kvstore='local')``: train_yolov3.py:603-606,527-530, detect_yolo3.py:211-213).  On the
MI355X node the unit is one process per GPU over RCCL, so a script that is started the reference's
way — ``python train.py --gpus 0,1,2,3`` — re-runs ITSELF as N rank processes:

    if launch.needs_spawn(n):            # WORLD_SIZE unset and n > 1
        sys.exit(launch.spawn_ranks(n))  # the parent only waits; it never initialises HIP

The children get RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT (a free
port) — the environment ``torch.distributed.run`` would set, so the same script also runs under
torchrun unchanged.  Nothing here ``exec``s: ranks are plain child processes and the parent returns
the first non-zero exit code (remaining ranks are terminated, so a crashed rank cannot leave the
others hanging in a collective).
"""
import os
import socket
import subprocess
import sys
import time


def needs_spawn(n_ranks):
    """True in a parent that was started without a rank environment and wants more than one rank."""
    return n_ranks > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_env(rank, world, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // world)))
    return env


def spawn_ranks(n_ranks, argv=None, timeout=None, poll=0.05):
    """Run ``sys.executable argv`` as n_ranks rank processes and wait.  Rank 0 inherits stdout (its
    JSON line is the parent's output); every rank inherits stderr.  Returns the exit code."""
    argv = list(sys.argv if argv is None else argv)
    port = free_port()
    procs = []
    for r in range(n_ranks):
        procs.append(subprocess.Popen([sys.executable] + argv, env=rank_env(r, n_ranks, port),
                                      stdout=None if r == 0 else subprocess.DEVNULL))
        if os.environ.get("VY_LAUNCH_TRACE"):
            sys.stderr.write("[launch] rank %d/%d pid %d port %d\n" % (r, n_ranks, procs[-1].pid, port))
    t0 = time.time()
    rc = 0
    live = list(procs)
    try:
        while live:
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code
            if rc != 0 or (timeout is not None and time.time() - t0 > timeout):
                if rc == 0:
                    rc = 124
                break
            time.sleep(poll)
    finally:
        for p in live:  # the exact PIDs this parent started
            p.terminate()
        for p in live:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    return rc if rc >= 0 else 128 - rc
