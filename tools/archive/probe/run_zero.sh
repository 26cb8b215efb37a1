P=tools/probe/conv_split_probe
for i in 1 2; do
timeout 300 $P 64 76 128 256 3 1 0 200 | grep "^conv" | tail -1 | cut -c60-
VY_PROBE_ZERO=1 timeout 300 $P 64 76 128 256 3 1 0 200 | grep "^conv" | tail -1 | cut -c60-
done
