import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import iif_amd_loader; iif = iif_amd_loader.load()
import bench
for th in (8, 32, 64, 128, 256):
    t = time.time(); v, secs, m = bench.cpu_baseline(iif, 300, 200, th); print(th, 'threads:', round(v, 1), 'msg/s solve', round(secs, 2), 's, total', round(time.time() - t, 1), flush=True)
