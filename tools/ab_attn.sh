for b in 7 8 14; do echo "B=$b"; B=$b python tools/run_attn.py 2>&1 | grep -E "attn_fwd|single pass \(no G\)"; done
