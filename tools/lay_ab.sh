R=$GRAFT_REPO_ROOT
cd /tmp
for v in prev new rule2; do
cp $R/videoyolo_amd/libvyolo_$v.so $R/videoyolo_amd/libvyolo.so
python3 $R/tools/layer_profile.py --size 608 --batch 64 --out $R/gpurun_out/r06_lay_$v.txt > /dev/null 2>&1
done
cp $R/videoyolo_amd/libvyolo_new.so $R/videoyolo_amd/libvyolo.so
