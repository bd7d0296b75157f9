"""Generates tests/golden/tokenizer_ids.json by IMPORTING the reference's tokenizer (the only reference module that
imports in the build container: it needs just `tokenizers`).  Run in the build container only:
    python tests/golden/make_tokenizer_fixture.py
The output is data (inputs + expected ids); nothing of the reference's source is stored."""
import json
import os
import sys

sys.path.insert(0, '/root/reference')
from mreserve import lowercase_encoder as le  # noqa: E402

enc = le.get_encoder()
out = {'vocab_size': enc.get_vocab_size(),
       'special_ids': {k: getattr(le, k) for k in ['PADDING', 'START', 'END', 'MASK', 'MASKAUDIO', 'AUDIOSPAN', 'LTOVPOOL', 'RESETCTX']},
       'special_tokens': {t: enc.token_to_id(t) for t in ['<|PAD|>', '<|START|>', '<|END|>', '<|MASK|>', '<|MASKAUDIO|>',
                                                          '<|AUDIOSPAN|>', '<|LTOVPOOL|>']},
       'encode': {s: enc.encode(s).ids for s in ["in this video i'll be<|MASK|>", 'answer: ', 'rationale: ', 'title:',
                                                  'description:', 'tags:']}}
enc.enable_padding(pad_token='<|PAD|>', length=15)
out['encode_padded_15'] = {s: enc.encode(s).ids[:15] for s in ['making coffee', 'going backpacking']}
enc.no_padding()
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tokenizer_ids.json'), 'w') as f:
    json.dump(out, f, indent=1, sort_keys=True)
print(json.dumps(out)[:400])
