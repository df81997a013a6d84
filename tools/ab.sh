for i in 1 2 3; do
RENE_HIP_LIB=librene_hip_notos.so timeout -k 10 100 python3 tools/dev.py rate dragon-class teapot-class --launches 16 2>&1 | grep Mrays | sed 's/^/base /'
timeout -k 10 100 python3 tools/dev.py rate dragon-class teapot-class --launches 16 2>&1 | grep Mrays | sed 's/^/new  /'
done
