for t in "" "attn_simple_db=0" "attn_simple_remap=1" "attn_res=2" "attn_reg=2"; do
  echo "== $t"; SOLA_TUNE="$t" python tools/ragged_infer_time.py
done
