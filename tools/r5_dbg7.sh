L=dualsuperreslearningforsemseg_amd/libdsrl_hip.so
cp $L /tmp/new.so
cp ab/libdsrl_hip_prev.so $L; echo "== prev"; timeout -k 10 200 python tools/r5_dbg7.py 2>&1 | grep -v amdgpu
cp /tmp/new.so $L; echo "== new"; timeout -k 10 200 python tools/r5_dbg7.py 2>&1 | grep -v amdgpu
