# sourced at the top of a gpurun command line: scratch dirs and the rocprofv3 working-directory rule
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
