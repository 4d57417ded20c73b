for v in stg2 stg5; do bash tools/r06_ab_generic.sh $v 2>&1 | grep -v "^cfg2 lib= \|^serial lib= \|^cfg4 lib=:" ; done
bash tools/r06_ab_generic.sh none 2>&1 | grep "lib= \|lib=:"
