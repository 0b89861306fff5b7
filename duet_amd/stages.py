# coding=utf-8
"""Steps B, C, D of the pipeline: thin shells around external tools (Clair3, cuteSV / Sniffles / SVIM,
bcftools, WhatsHap).  They are not part of the accelerated path; they exist so that the `duet`
command keeps working end to end where those tools are installed.  Command lines are the ones
upstream builds (snp_calling.py:13-18, sv_calling.py:13-20, snp_phasing.py:17-32); exit codes are
ignored as upstream does.
"""

import logging
import os
import shutil
import time

_BAR = '*' * 25


def _stage(name):
    def wrap(fn):
        def run(*a, **kw):
            logging.info('%s %s STARTED %s' % (_BAR, name, _BAR))
            t0 = time.time()
            fn(*a, **kw)
            logging.info('%s %s COMPLETED IN %ss %s' % (_BAR, name, round(time.time() - t0, 3), _BAR))
        run.__name__ = fn.__name__
        return run
    return wrap


def _sh(cmd):
    tool = cmd.split()[0]
    if tool not in ('mkdir', 'bash', 'chmod') and shutil.which(tool) is None:
        logging.warning('external tool not found on PATH: ' + tool)
    os.system(cmd)


@_stage('SNP CALLING')
def snp_calling(home, ref_path, aln_path, min_af, thread, include_all_ctgs):
    out = home + '/snp_calling/'
    _sh('mkdir ' + out)
    cmd = ['run_clair3.sh', '-b', aln_path, '-f', ref_path, '-m', '"${CONDA_PREFIX}/bin/models/ont"',
           '-t', str(thread), '-p', 'ont', '-o', out, '--snp_min_af=' + str(min_af), '--pileup_only',
           '--call_snp_only']
    if include_all_ctgs:
        cmd.append('--include_all_ctgs')
    _sh(' '.join(cmd))


@_stage('SV CALLING')
def sv_calling(home, ref_path, aln_path, cls_thres, svlen_thres, thread, caller, supp_thres):
    out = home + '/sv_calling/'
    _sh('mkdir ' + out)
    if caller == 'svim':
        _sh(' '.join(['svim alignment', out, aln_path, ref_path, '--min_sv_size', str(svlen_thres), '--read_names',
                      '--minimum_depth 0 --minimum_score 0 --cluster_max_distance', str(cls_thres)]))
    elif caller == 'cutesv':
        _sh(' '.join(['cuteSV --genotype --report_readid', aln_path, ref_path, out + 'variants.vcf', out,
                      '-t', str(thread), '-s', str(supp_thres), '-l', str(svlen_thres)]))
    elif caller == 'sniffles':
        _sh(' '.join(['sniffles --input', aln_path, '--vcf', out + 'variants.vcf', '-t', str(thread),
                      '--output-rnames --allow-overwrite']))


@_stage('SNP PHASING')
def snp_phasing(home, ref_path, aln_path, thread):
    import shlex
    import subprocess
    out = home + '/snp_phasing/'
    pile = home + '/snp_calling/pileup.vcf.gz'
    _sh('mkdir ' + out)
    ctgs = subprocess.check_output(shlex.split('tabix --list-chroms ' + pile)).decode('ascii').split('\n')[:-1]
    for c in ctgs:
        _sh('bcftools view -r ' + c + ' -c1 ' + pile + ' > ' + out + c + '.vcf')
    par = 'parallel -j' + str(thread) + ' "'
    each = '" ::: ${CHR[@]}\n'
    script = out + 'parallel_wh.sh'
    with open(script, 'w') as fh:
        fh.write('CHR=(' + ' '.join(ctgs) + ')\n')
        fh.write(par + 'whatshap phase -o ' + out + 'phased_{1}.vcf.gz -r ' + ref_path + ' --chromosome {1} '
                 '--distrust-genotypes --ignore-read-groups ' + out + '{1}.vcf ' + aln_path + each)
        fh.write(par + 'tabix -f -p vcf ' + out + 'phased_{1}.vcf.gz' + each)
        fh.write(par + 'whatshap haplotag -o ' + out + '{1}.bam -r ' + ref_path + ' --regions {1} '
                 '--ignore-read-groups --tag-supplementary ' + out + 'phased_{1}.vcf.gz ' + aln_path + each)
    _sh('chmod a+x ' + script)
    _sh('bash ' + script)
