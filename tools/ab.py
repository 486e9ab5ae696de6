#!/usr/bin/env python3
"""A/B timing of builds of the library in one GPU session: for every build and workload the
kernel time of a few 64-spp launches of the full frame (HIP events), best and mean.
usage: ab.py [--workloads c2,c4] [--launches 5] [--checksum] libA.so libB.so ...
With --checksum the packed frame's CRC is printed too (identical pictures across builds)."""
import argparse
import os
import subprocess
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(lib, workloads, launches, checksum, sps_max=64):
    os.environ["RT_HIP_LIBRARY"] = os.path.abspath(lib)
    sys.path.insert(0, ROOT)
    from raytracing_amd import Renderer, load_tables, make_params, workloads as wls
    tables = load_tables()
    for name in workloads:
        wl = wls.workload(name)
        w, h, spp = wl["width"], wl["height"], wl["spp"]
        sps = min(sps_max, spp)
        with Renderer(0, tables) as r:
            r.set_scene(wl["flat"]); r.set_camera(wl["camera"]); r.begin_frame(w, h)
            r.render(make_params(w, h, spp=spp, sample_begin=0, sample_count=sps)); r.sync()
            r.begin_frame(w, h)
            seg0 = r.ray_count()
            ms = []
            for k in range(launches):
                r.timer_start()
                r.render(make_params(w, h, spp=spp, sample_begin=0, sample_count=sps))
                ms.append(r.timer_stop())
            seg = (r.ray_count() - seg0) / launches
            crc = zlib.crc32(r.read_packed().tobytes()) if checksum else 0
        print("%-4s %-44s best %8.3f ms  mean %8.3f ms  %9.1f Mrays/s%s"
              % (name, os.path.basename(lib), min(ms), sum(ms) / len(ms), seg / min(ms) / 1e3,
                 "  crc %08x" % crc if checksum else ""), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="c2,c4")
    ap.add_argument("--launches", type=int, default=5)
    ap.add_argument("--checksum", action="store_true")
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--sps", type=int, default=64, help="samples per rt_render call")
    ap.add_argument("libs", nargs="+")
    a = ap.parse_args()
    wl = [x for x in a.workloads.split(",") if x]
    if a.child:
        child(a.libs[0], wl, a.launches, a.checksum, a.sps)
        return
    for lib in a.libs:
        cmd = [sys.executable, os.path.abspath(__file__), "--child", "--workloads", a.workloads,
               "--launches", str(a.launches), "--sps", str(a.sps), lib] + (["--checksum"] if a.checksum else [])
        rc = subprocess.run(cmd, timeout=400).returncode
        if rc != 0:
            sys.exit(rc)


if __name__ == "__main__":
    main()
