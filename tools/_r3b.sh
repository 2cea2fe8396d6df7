set -e
bash tools/pmc_quick.sh path_tracer_amd/libptmi.so new
bash tools/pmc_quick.sh build/variants/r2.so r2
