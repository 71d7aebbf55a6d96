cp fusion4landslide_amd/lib/libf4l_hip.so /tmp/keep.so
cp tools/gpu/ab/prof.so fusion4landslide_amd/lib/libf4l_hip.so
python tools/gpu/wg_bench.py 2>&1 | grep -v "^\[icp prof\]" | tail -16
cp /tmp/keep.so fusion4landslide_amd/lib/libf4l_hip.so
