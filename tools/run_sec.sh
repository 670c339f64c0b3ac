python tools/exp_sector.py 12 5 --sector-only 2>&1 | grep -E "^sector=1|^E " | cut -c1-100
python tools/exp_sector.py 13 6 --sector-only| cut -c1-100
python tools/exp_sector.py 13 6 --sector-only --opt=sector_threads=512| cut -c1-100
python tools/exp_sector.py 13 6 --sector-only --opt=sector_threads=256| cut -c1-100
