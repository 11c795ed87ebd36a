#!/bin/bash
# One bench line per configuration (BASELINE.json configs[1], [2], cascade, fp16, HIP graph, C5) under gpurun_out/<tag>_bench_line*.json
tag=$1
o=gpurun_out/${tag}_bench_line
python bench.py > ${o}.json 2> ${o}.err
python bench.py --model transeg > ${o}_transeg.json 2>> ${o}.err
python bench.py --model cascade --no-cpu-baseline > ${o}_cascade.json 2>> ${o}.err
python bench.py --model cascade --no-cpu-baseline --seg-mode same > ${o}_cascade_segsame.json 2>> ${o}.err
python bench.py --dtype fp16 > ${o}_fp16.json 2>> ${o}.err
python bench.py --graph --no-cpu-baseline --no-fp32-leg > ${o}_graph.json 2>> ${o}.err
python bench.py --model cascade --dtype fp16 --size 192 192 128 --batch 1 --roi 96 --checkpoint --loss-scale 1024 --no-cpu-baseline > ${o}_c5.json 2>> ${o}.err
python bench.py --model cascade --dtype fp16 --size 192 192 128 --batch 1 --roi 96 --checkpoint --loss-scale 1024 --no-cpu-baseline --seg-mode same > ${o}_c5_segsame.json 2>> ${o}.err
python bench.py --dtype fp32x3 --no-cpu-baseline > ${o}_fp32x3.json 2>> ${o}.err
# OAR-TRANSEG at the reference's own training crop (OARSegmentation/config.py:24: 96^3), four crops per step
python bench.py --model transeg --size 96 --batch 4 --no-cpu-baseline > ${o}_transeg_crop96.json 2>> ${o}.err
python bench.py --model transeg --size 96 --batch 4 --dtype fp32x3 --no-cpu-baseline > ${o}_transeg_crop96_fp32x3.json 2>> ${o}.err
for f in ${o}*.json; do echo "$f: $(python tools/show_bench.py $f 2>/dev/null | head -1)"; done
tail -5 ${o}.err
