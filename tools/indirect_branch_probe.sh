# Compiles tools/indirect_branch_probe.hip three ways for gfx950 and reports whether an indirect jump made it into the ISA.
cd "$(dirname "$0")"
for p in 1 2 3; do
	if /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 --cuda-device-only -DPROBE=$p -S -o /tmp/ibp_$p.s indirect_branch_probe.hip > /tmp/ibp_$p.log 2>&1; then
		echo "PROBE=$p: compiled; s_setpc_b64 in the ISA: $(grep -c s_setpc_b64 /tmp/ibp_$p.s); flag-register branches (s_cbranch_vcc*): $(grep -c 's_cbranch_vcc' /tmp/ibp_$p.s)"
	else
		echo "PROBE=$p: compiler failed: $(grep -m1 -o "Running pass '[^']*' on function" /tmp/ibp_$p.log | tail -1)"
	fi
done
