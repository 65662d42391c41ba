cd /root/repo
for v in 1; do
echo "PC8=$v"
GAMMA_HIP_PROD_C8=$v timeout 300 python bench.py --cpu-seconds 0 --steps 40 2>&1 | grep "scan phases\|pc8:" | tail -2 | cut -c1-700
done
