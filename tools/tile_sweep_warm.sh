run() { local name=$1 custom=$2; shift 2
  for v in fast "$@"; do
    RAGGED=auto RES_SEP=1 CUSTOM="$custom" timeout 200 tools/bin/cb_$v 2>&1 | grep TF | awk -v v=$v -v n=$name '{print n, v, $2, $4, $(NF-3), "ms", $(NF-1), "TF"}'
  done; }
run s1 "32,256,3,1,3968;32,256,7,3,3968;32,256,11,5,3968" cfg0 cfg1 cfg2 cfg3 cfg6 cfg7
run s2 "32,128,3,1,31744;32,128,7,3,31744;32,128,11,5,31744" cfg0 cfg1 cfg2 cfg3 cfg6 cfg7
run s3 "32,64,3,1,63488;32,64,7,3,63488;32,64,11,5,63488" cfg2 cfg3 cfg6
