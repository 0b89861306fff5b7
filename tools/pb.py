import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('cfg2 ms/step',d['ms_per_step'],'iso',d['kernels_us_isolated'],'frac',d['roofline']['frac'])
e=d.get('extra',{})
if 'config3_1gpu_2e7_marks' in e:
    c=e['config3_1gpu_2e7_marks']; print('2e7 ms/step',c['ms_per_step'],c['kernels_ms'],c['classify_frac_of_8TBs'],c['parity_vs_oracle'])
