# text-encoder GPU validation: the new op tests + model tests, then the whole GPU suite if time allows
timeout 300 python -m pytest tests -m gpu -x -q -k "get_rows or masked or clip_text or t5_encoder or tokens_to" 2>&1 | tail -15
timeout 200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
