"""BASELINE.json configs[0] (CPU only, plumbing): the reference's criterion workloads
(plonky2/benches/field_arithmetic.rs, ffts.rs) over the C restatement — builds and runs oracle/bench_c1,
adds the host description. ns per iteration, one thread. usage: python tests/bench_c1.py [out.json]"""
import json
import os
import platform
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "bench_c1"], stdout=subprocess.DEVNULL)
    res = json.loads(subprocess.check_output([os.path.join(ROOT, "oracle", "bench_c1")]))
    cpu = ""
    try:
        cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    out = dict(config="configs[0]: field_arithmetic + forward NTT / rate-8 LDE at 2^13..2^16, CPU restatement (kind: port), 1 thread",
               unit="ns per iteration", cpu=cpu, logical_cpus=os.cpu_count(), machine=platform.machine(), results=res)
    text = json.dumps(out, indent=1)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
