"""-m "not gpu": the stream-K work split of the conv kernel (videoyolo_amd/csrc/sk_schedule.h — the header the HIP kernel
itself includes) compiled with g++ and run over the product's launch geometries and 20 000 random ones: every k-step of
every tile exactly once and in chain order, hand-offs only between neighbouring blocks of one XCD group, balanced shares
(tests/sk_schedule_check.cpp has the list)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_stream_k_schedule_covers_every_k_step_once(tmp_path):
    exe = str(tmp_path / "sk_schedule_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", "-o", exe, os.path.join(HERE, "sk_schedule_check.cpp")],
                   check=True)
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:]
    assert p.stdout.startswith("ok ")
