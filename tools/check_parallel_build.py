"""Developer check: the multi-threaded BVH build gives the tree of the single-threaded one (statistics, checksum, hits)."""
import os, sys, time, subprocess, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
if len(sys.argv) > 1:
    import numpy as np
    import helpers as H
    from pathtracer_amd import scenes
    model = scenes.readme_standin(detail=1.0) if sys.argv[1] == "readme" else scenes.stress_mesh(3_000_000)
    t = time.time()
    g = H.make_gpu(model, None, H.settings(program=0), 320, 180)
    load = time.time() - t
    infos = [g.bvh_info(i) for i in range(len(model.meshes))]
    hits = g.render_hits(320, 180)
    print(json.dumps({"load_s": load, "checksums": [int(i.checksum) for i in infos], "nodes": [int(i.n_nodes) for i in infos],
                      "hit_digest": int(np.frombuffer(hits.tobytes(), np.uint32).astype(np.uint64).sum())}))
    sys.exit(0)
for scene in ("readme", "stress"):
    res = {}
    for threads in ("1", "16"):
        env = dict(os.environ, PTGPU_BUILD_THREADS=threads)
        out = subprocess.run([sys.executable, __file__, scene], env=env, capture_output=True, text=True, timeout=600)
        res[threads] = json.loads(out.stdout.strip().splitlines()[-1])
    same = all(res["1"][k] == res["16"][k] for k in ("checksums", "nodes", "hit_digest"))
    print(scene, "identical:", same, "load 1 thread %.2f s, 16 threads %.2f s" % (res["1"]["load_s"], res["16"]["load_s"]))
    assert same
