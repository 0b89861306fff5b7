#!/usr/bin/env python3
# coding=utf-8
"""Round-3 golden fixtures from the UNMODIFIED reference (development container only; harness of make_golden.py: the
reference's own modules imported from /root/reference/src, `samtools view` a PATH shim that prints the text file laid
beside the path it is given).  Pins BASELINE configs[3] / [4] at their real SHAPE -- the 24-contig, 2e7-mark genome --
in the dialects those configurations produce:

    seeded_r3.json   per case: sha256 of the regenerated inputs and of the reference's phased_sv.vcf
                     * config3_svim      READS= / GT:DP:AD        (configs[3]: --sv_caller svim), -s 50 -r 2
                     * config3_sniffles  RNAMES= / GT:GQ:DR:DV    (configs[4]: --sv_caller sniffles, min_support_read=2;
                                         refread = GQ, quirk Q4 of SURVEY 8a)

One reference run each, ~5-10 min / ~12 GB.

    python tests/golden/make_golden_r3.py [svim] [sniffles]
"""

import json
import os
import shutil
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

from duet_amd import synth  # noqa: E402
import make_golden as G     # noqa: E402

N_MARKS = 20000000
SEED = 3


def main():
    dialects = [a for a in sys.argv[1:] if a in ('svim', 'sniffles')] or ['svim', 'sniffles']
    if not os.path.isdir(G.REF_SRC):
        sys.exit('reference not present: this script only runs in the development container')
    sys.path.insert(0, G.REF_SRC)
    tmp = tempfile.mkdtemp(prefix='duet_golden_r3_')
    G.install_shims(tmp)
    path = os.path.join(HERE, 'seeded_r3.json')
    seeded = json.load(open(path)) if os.path.exists(path) else []
    for dialect in dialects:
        home = os.path.join(tmp, 'config3_' + dialect)
        t0 = time.time()
        synth.write_workdir(home, synth.bench_genome(N_MARKS, SEED), dialect=dialect, seed=SEED, write_bam=False)
        print('config 3 (%s) text written in %.0f s' % (dialect, time.time() - t0), flush=True)
        t0 = time.time()
        G.run_reference(home, 50, 2)
        out = os.path.join(home, 'phased_sv.vcf')
        nrows = sum(1 for l in open(out) if not l.startswith('#'))
        seeded = [s for s in seeded if not (s['kind'] == 'config3' and s['dialect'] == dialect)]
        seeded.append(dict(kind='config3', seed=SEED, dialect=dialect, svlen_thres=50, suppread_thres=2, all_ctgs=False,
                           inputs_sha256=G.inputs_digest(home), output_sha256=G.sha256_file(out), rows=nrows,
                           reference_seconds=round(time.time() - t0, 2)))
        print('seeded config3/%s: %d rows in %.1f s' % (dialect, nrows, time.time() - t0), flush=True)
        shutil.rmtree(home)
        with open(path, 'w') as f:
            json.dump(seeded, f, indent=1)
    shutil.rmtree(tmp)


if __name__ == '__main__':
    main()
