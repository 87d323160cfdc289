for pad in 0 12000 28000; do echo "a4w4 (no twiddles) pad=$pad"; SP_CARRY_LDS_PAD=$pad SP_LIB_PATH=build/variants/a4w4/libspectral.so timeout -k 10 100 python tools/kbench.py 2>&1 | grep welch; done
for pad in 0 8000; do echo "main (LTW+WLDS, 3 waves) pad=$pad"; SP_CARRY_LDS_PAD=$pad timeout -k 10 100 python tools/kbench.py 2>&1 | grep welch; done
for pad in 0 12000 28000; do echo "noltw pad=$pad"; SP_CARRY_LDS_PAD=$pad SP_LIB_PATH=build/variants/noltw/libspectral.so timeout -k 10 100 python tools/kbench.py 2>&1 | grep welch; done
