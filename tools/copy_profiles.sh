# copies the judged summaries of gpurun_out/prof_<tag> (tools/collect_profiles.sh) into profiles/<tag>_*
TAG=${1:-r02}
O=gpurun_out/prof_$TAG
cp $O/bench_n1_fp16_fp8.json profiles/${TAG}_bench_n1_fp16_fp8.json
cp $O/bench_n1_fp16x3.json profiles/${TAG}_bench_n1_fp16x3.json
[ -f $O/bench_n1_fp16_e4m3.json ] && cp $O/bench_n1_fp16_e4m3.json profiles/${TAG}_bench_n1_fp16_e4m3.json
[ -f $O/bench_n1_fp16x3_asm.json ] && cp $O/bench_n1_fp16x3_asm.json profiles/${TAG}_bench_n1_fp16x3_asm.json
for n in SQ_WAVE_CYCLES GRBM_GUI_ACTIVE; do [ -f $O/pmcx_$n.txt ] && cp $O/pmcx_$n.txt profiles/${TAG}_pmcx_$n.txt; done
for n in SQ_LDS_IDX_ACTIVE; do [ -f $O/pmc_$n.txt ] && cp $O/pmc_$n.txt profiles/${TAG}_pmc_$n.txt; done
for n in SQ_WAVE_CYCLES GRBM_GUI_ACTIVE; do [ -f $O/pmc8_$n.txt ] && cp $O/pmc8_$n.txt profiles/${TAG}_pmc8_$n.txt; done
cp $(ls -t $O/trace/*/*kernel_stats.csv | head -1) profiles/${TAG}_kernel_stats.csv      # the newest: gpurun merges runs into one directory
for n in FETCH_SIZE GRBM_GUI_ACTIVE SQ_WAVE_CYCLES TCC_HIT_sum WRITE_SIZE; do cp $O/pmc_$n.txt profiles/${TAG}_pmc_$n.txt; done
cp $O/traffic.json profiles/${TAG}_traffic.json
ls $O/trace_trained/*/*kernel_stats.csv > /dev/null 2>&1 && cp $(ls -t $O/trace_trained/*/*kernel_stats.csv | head -1) profiles/${TAG}_trained_kernel_stats.csv
grep -h "^stress\|MISMATCH" $O/pmc_stress.log > profiles/${TAG}_pmc_stress.txt
cp $(ls -t $O/teacher_trace/*/*kernel_stats.csv | head -1) profiles/${TAG}_teacher_kernel_stats.csv
grep teacher $O/teacher_time.txt > profiles/${TAG}_teacher_time.txt
cp $O/teacher_pmc_SQ_WAVE_CYCLES.txt profiles/${TAG}_teacher_pmc_SQ_WAVE_CYCLES.txt
cp $O/teacher_pmc_GRBM_GUI_ACTIVE.txt profiles/${TAG}_teacher_pmc_GRBM_GUI_ACTIVE.txt
python - <<PY
import re
src = open('$O/power.txt').read()
out = ['rocm-smi while tools/body_time.py renders 2,500 frames back to back (fp16_fp8, 800x800; tools/power_sample.sh):',
       'package power cap (rocm-smi --showmaxpower): 1400 W']
for l in src.split('\n'):
    m = re.search(r'sclk clock level: \S+ \((\d+)Mhz\).*Power \(W\): ([\d.]+)', l)
    if m:
        out.append('  sclk %s MHz   package power %s W' % (m.group(1), m.group(2)))
out.append([l for l in src.split('\n') if 'frame median' in l][-1])
open('profiles/${TAG}_power.txt', 'w').write('\n'.join(out) + '\n')
PY
python tools/kernel_resources.py $TAG > /dev/null
