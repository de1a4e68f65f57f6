#!/bin/bash
# Exports the shipped Keras weights of the Kessler surrogate (104 fp32 parameters) and the min/max scaling
# tables to plain text, because neither h5py nor an HDF5 runtime can be assumed on the GPU box.
# Run in the build container only (needs /root/reference and /opt/conda/bin/h5dump).  %.9g round-trips fp32.
#   source: experiments/supercell_kessler_surrogate/inputs/examples/supercell_kessler_singlecell_model_weights.h5
#           datasets /dense_6/dense_6/{kernel:0 (5,10), bias:0 (10)}, /dense_7/dense_7/{kernel:0 (10,4), bias:0 (4)}
#           (microphysics_kessler_ponni.h:103-107)
set -e
SRC=/root/reference/experiments/supercell_kessler_surrogate/inputs/examples
OUT=$(dirname "$0")/../miniweatherml_amd/data
H5=$SRC/supercell_kessler_singlecell_model_weights.h5
dump() { /opt/conda/bin/h5dump -m "%.9g" -d "$1" "$H5" | grep -E '^\s*\(' | sed -E 's/^\s*\([0-9,]+\):\s*//; s/,\s*$//' | tr ',' '\n' | sed 's/ //g' | grep -v '^$'; }
{
  echo "# dense_6 kernel (5,10) row-major"; dump "/dense_6/dense_6/kernel:0"
  echo "# dense_6 bias (10)";               dump "/dense_6/dense_6/bias:0"
  echo "# dense_7 kernel (10,4) row-major"; dump "/dense_7/dense_7/kernel:0"
  echo "# dense_7 bias (4)";                dump "/dense_7/dense_7/bias:0"
} > $OUT/kessler_surrogate_weights.txt
cp $SRC/supercell_kessler_stencil_input_scaling.txt  $OUT/kessler_surrogate_input_scaling.txt
cp $SRC/supercell_kessler_stencil_output_scaling.txt $OUT/kessler_surrogate_output_scaling.txt
grep -vc '^#' $OUT/kessler_surrogate_weights.txt
