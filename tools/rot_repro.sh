for p in oneway pingpong_tail zigzag; do echo "=== $p"; timeout 60 python tools/rot_patterns.py $p 2>&1 | grep -v amdgpu.ids | tail -4; done
