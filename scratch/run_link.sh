timeout 600 python tools/big_linkage.py 86373 64,128 256 2 2>&1 | tail -2
timeout 600 python tools/big_linkage.py 43173 64,128 256 2 2>&1 | tail -2
timeout 600 python tools/big_linkage.py 10000 32,64 256 2 2>&1 | tail -2
timeout 600 python tools/big_linkage.py 5000 0,16,32,64 256 2 2>&1 | tail -4
timeout 600 python tools/big_linkage.py 2500 0,16,32 256 2 2>&1 | tail -3
