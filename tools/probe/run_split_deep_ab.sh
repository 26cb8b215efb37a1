P=tools/probe/conv_split_probe
for B in 1 2 4; do
  for shape in "19 512 1024 3" "38 256 512 3" "76 128 256 3" "152 64 128 3" "19 1024 512 1" "38 512 256 1"; do
    echo "--- B=$B $shape"
    VY_SPLIT_DEEP=0 timeout 120 $P $B $shape 1 0 30 | grep -E "^\[" | tail -1 | cut -c1-14,100-
    timeout 120 $P $B $shape 1 0 30 | grep -E "^\[|float64" | tail -2 | cut -c1-14,100-
  done
done
