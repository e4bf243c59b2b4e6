#!/bin/bash
export TMPDIR=/tmp
for n in none decoder_token_chain cost_lookup9x9 flow_encode narrow layernorm window_attention attention_kvlds attention_small latent_pool sine_pe gma_aggregate sepconv_gru linear_chain128 none; do
  python tools/ablate.py $n 2>/dev/null | grep ablate
done
