run() { # name ragged custom cfgs...
  local name=$1 rg=$2 custom=$3; shift 3
  for v in base "$@"; do
    RAGGED=$rg CUSTOM="$custom" timeout 120 tools/bin/cb_$v 2>&1 | grep TF | awk -v v=$v -v n=$name '{print n, v, $2, $4, $(NF-3), "ms", $(NF-1), "TF"}'
  done
}
run s2 64 "32,128,3,1,31744;32,128,7,3,31744;32,128,11,5,31744" cfg0 cfg1 cfg2 cfg3 cfg6 cfg7
run s3 128 "32,64,3,1,63488;32,64,7,3,63488;32,64,11,5,63488" cfg2 cfg3 cfg6
run s4 256 "32,32,3,1,126976;32,32,7,3,126976;32,32,11,5,126976" cfg4 cfg5
