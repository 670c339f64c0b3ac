for args in "" "--opt=sector_bits=15" "--opt=sector_bits=14" "--opt=sector_bits=14 --opt=sector_threads=64"; do
echo "== $args"; python tools/exp_sector.py 12 5 $args 2>&1 | grep -E "^sector=1|^E " | cut -c1-100
done
