#!/bin/bash
# usage: bash profiles/variants.sh "<probe command>" <variant>...  -- runs the probe against build_diag/<variant>/libmvk.so ("base": the in-tree library)
CMD=$1; shift
for v in "$@"; do
  echo "== $v"
  if [ "$v" = base ]; then timeout -k 10 150 $CMD || exit 1
  else MVK_LIB=build_diag/$v/libmvk.so timeout -k 10 150 $CMD || exit 1; fi
done
