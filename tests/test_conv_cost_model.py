"""-m "not gpu": the conv launch cost model (videoyolo_amd/csrc/conv_cost_model.h, the header conv_igemm.hip includes)
compiled with g++: the (block tile, stream-K?) decisions for BASELINE's conv shapes as they were measured and adopted on
the MI355X (profiles/r03_layers_608_b64.txt, r03_train416_b16_layers.txt), and the model's structural properties."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))

# M N K -> tile of the exact kernel | conv mode split_bf16x3: the kernel the launch goes to (round 4;
# profiles/r04_layers_608_b64_split.txt, tools/probe/run_split_batch_sweep.sh, run_split_ksplit_sweep.sh):
# 608x608 batch 64, then 416x416 batch 16 (training forward shapes), then batch 1; third column: 3x3 cells the Winograd
# F(2, 3) instance takes instead (conv_wino.hip, vy_predict_wino; fitted on tools/layer_profile.py sweeps).
# After "-- 128 CUs": the exact kernel's tiles on a device of 128 CUs (the models count rounds in the device's CU count;
# the split / Winograd choices are refused there: vy_model_fitted, checked by the program itself)
EXPECTED = """\
5914624 64 288 -> 128x64   | split 256x64
5914624 32 64 -> 128x32
1478656 128 576 -> 128x128   | split 128x128   | wino
1478656 64 128 -> 128x64   | split 128x64
369664 256 1152 -> 128x128   | split 128x128   | wino
369664 128 256 -> 128x128   | split 128x128
92416 512 2304 -> 128x128sk   | split 128x128   | wino
92416 256 512 -> 128x128   | split 128x128
23104 1024 4608 -> 128x128sk   | split 128x128   | wino
23104 512 1024 -> 128x128   | split 128x128
23104 75 1024 -> 64x64
43264 256 1152 -> 128x128sk   | split 128x128   | -
10816 512 2304 -> 128x64sk   | split 128x128 k2   | -
2704 1024 4608 -> 64x64sk   | split 128x64 k2   | -
2704 512 1024 -> 64x64   | split 128x128 k5
43264 128 256 -> 128x64   | split 128x64
5776 256 1152 -> 64x64   | split 128x128 k5   | -
361 1024 4608 -> 64x64   | split 128x128 k10   | -
-- 128 CUs
5914624 64 288 -> 128x64
5914624 32 64 -> 128x32
1478656 128 576 -> 128x128
1478656 64 128 -> 128x64
369664 256 1152 -> 128x128
369664 128 256 -> 128x128
92416 512 2304 -> 128x128
92416 256 512 -> 128x64
23104 1024 4608 -> 128x64
23104 512 1024 -> 128x128
23104 75 1024 -> 128x64
43264 256 1152 -> 128x64
10816 512 2304 -> 64x64
2704 1024 4608 -> 128x64
2704 512 1024 -> 64x64
43264 128 256 -> 128x128
5776 256 1152 -> 64x64
361 1024 4608 -> 64x64
"""


def test_tile_and_stream_k_decisions(tmp_path):
    exe = str(tmp_path / "conv_cost_model_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", "-o", exe,
                    os.path.join(HERE, "conv_cost_model_check.cpp")], check=True)
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=60)
    assert p.returncode == 0, p.stderr
    assert p.stdout == EXPECTED, p.stdout
