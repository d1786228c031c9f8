for L in "" tools/exp/libnbp_w5.so tools/exp/libnbp_w6.so; do
  echo "== lib ${L:-default}"
  NBP_LIB_OVERRIDE=$L python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys; sys.path.insert(0, "tools/exp"); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import lcv_f32_check as c
from parity_utils import abi
for N, nf, man in ((200, 8192, abi.EUCLID2), (200, 2048, abi.EUCLID2), (300, 4096, abi.EUCLID3), (200, 8192, abi.SE2)):
    a = c.run(N, nf, man, "gauss", True); b = c.run(N, nf, man, "gauss", False)
    print(N, nf, man, "all-double %.3f ms  bracketed %.3f ms" % (a[0], b[0]), a[3] == b[3])
PY
done
