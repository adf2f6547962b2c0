TESTLIBS="" bash scripts/gpu_ab.sh r03_d tree rq4
timeout 900 python scripts/parity_report.py --cases cfg1,crop2,crop3 --out gpurun_out/r03_d/parity_table.json > gpurun_out/r03_d/parity.log 2>&1; tail -c 600 gpurun_out/r03_d/parity.log
timeout 700 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -15
