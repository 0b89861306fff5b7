#!/usr/bin/env python3
# coding=utf-8
"""Round-5 golden fixture from the UNMODIFIED reference (development container only; harness of make_golden.py: the
reference's own modules imported from /root/reference/src, `samtools view` a PATH shim).  Pins BASELINE configs[1] on SURVEY.md
section 8d's generator TO THE LETTER (duet_amd.synth.bench_contig(..., literal_8d=True): 20 % of the marks' names absent from the
tag tables, pc = floor(Exp(mean 600)); the bench's default generator -- 5 % absent, a geometric PC -- is pinned by seeded.json):

    seeded_r5.json   sha256 of the regenerated inputs and of the reference's phased_sv.vcf for kind `config2_8d`

    python tests/golden/make_golden_r5.py
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

from tests import helpers as H     # noqa: E402
import make_golden as G             # noqa: E402

SEED = 1


def main():
    if not os.path.isdir(G.REF_SRC):
        sys.exit('reference not present: this script only runs in the development container')
    sys.path.insert(0, G.REF_SRC)
    tmp = tempfile.mkdtemp(prefix='duet_golden_r5_')
    G.install_shims(tmp)
    home = os.path.join(tmp, 'config2_8d')
    contigs = H.build_case(home, 'config2_8d', SEED, 'cutesv', write_bam=False)
    t0 = time.time()
    G.run_reference(home, 50, 2)
    out = os.path.join(home, 'phased_sv.vcf')
    nrows = sum(1 for l in open(out) if not l.startswith('#'))
    rec = [dict(kind='config2_8d', seed=SEED, dialect='cutesv', svlen_thres=50, suppread_thres=2, all_ctgs=False,
                inputs_sha256=G.inputs_digest(home), output_sha256=G.sha256_file(out), rows=nrows,
                marks=int(contigs[0].cand_off[-1]), reference_seconds=round(time.time() - t0, 2))]
    print(rec[0])
    with open(os.path.join(HERE, 'seeded_r5.json'), 'w') as f:
        json.dump(rec, f, indent=1)
    shutil.rmtree(tmp)


if __name__ == '__main__':
    main()
