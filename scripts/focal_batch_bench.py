"""The two focal-length estimators through pl_estimate_batch (driver_focal_group.inc): problems per second at n correspondences,
40 % outliers, for several group sizes and worker counts; next to it the single-problem path from a pool of host threads.

    python scripts/focal_batch_bench.py [problems=1024] [n=2000] > profiles/r05_focal_batch.md
"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import poselib_amd as P  # noqa: E402
from poselib_amd import synth  # noqa: E402


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    da = [synth.absolute_pose_scene(n, 0.4, 7100 + k) for k in range(4)]
    dr = [synth.relative_pose_scene(n, 0.4, 7000 + k) for k in range(4)]

    def item(name, j):
        if name == "pnpf":
            d = da[j % 4]
            return ("abs", d["p2d"], d["p3d"], d["camera"], {"max_error": 4.0, "estimate_focal_length": True, "ransac": {"seed": j}})
        d = dr[j % 4]
        return ("shared_focal", d["x1"], d["x2"], d["camera1"]["params"][1:3], {"max_error": 2.0, "ransac": {"seed": j}})

    def single(name, j):
        it = item(name, j)
        if name == "pnpf":
            return P.estimate_absolute_pose(*it[1:])
        return P.estimate_shared_focal_relative_pose(*it[1:])

    print(f"# r05 - the focal-length estimators through pl_estimate_batch ({count} problems per call, {n} correspondences, 40 % outliers)\n")
    print("| estimator | path | group size | workers | problems/s |")
    print("|---|---|---|---|---|")
    for name in ("pnpf", "shared_focal"):
        # single-problem path, 16 host threads
        T, N = 16, max(256, count // 4)
        nxt, lock, bar = [0], threading.Lock(), threading.Barrier(T + 1)

        def work(i):
            single(name, i), single(name, i + 1)
            bar.wait()
            while True:
                with lock:
                    j = nxt[0]
                    nxt[0] += 1
                if j >= N:
                    break
                single(name, j)

        th = [threading.Thread(target=work, args=(i,)) for i in range(T)]
        for t in th:
            t.start()
        bar.wait()
        t0 = time.perf_counter()
        for t in th:
            t.join()
        print(f"| {name} | single-problem entry point, 16 host threads | - | - | {N / (time.perf_counter() - t0):.0f} |", flush=True)
        for group, workers in ((0, 8), (8, 8), (16, 8), (32, 8), (64, 8), (128, 8), (0, 4), (0, 16)):
            if group:
                os.environ["POSELIB_AMD_FOCAL_GROUP"] = str(group)
            else:
                os.environ.pop("POSELIB_AMD_FOCAL_GROUP", None)
            P.Batch([item(name, j) for j in range(count)]).run(workers)
            best = 0.0
            for _ in range(2):
                b = P.Batch([item(name, j) for j in range(count)])
                t0 = time.perf_counter()
                b.run(workers)
                best = max(best, count / (time.perf_counter() - t0))
            print(f"| {name} | pl_estimate_batch | {group or 'default'} | {workers} | {best:.0f} |", flush=True)
        os.environ.pop("POSELIB_AMD_FOCAL_GROUP", None)


if __name__ == "__main__":
    main()
