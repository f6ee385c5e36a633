for i in 1 2 3; do for v in prev cur; do echo "$v $(VETO_AMD_LIB=build/libveto_ffn_$v.so python tools/layer_tail_bench.py 2>&1 | grep 'one launch')"; done; done
