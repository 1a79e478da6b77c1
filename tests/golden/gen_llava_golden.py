"""Golden vectors for the LLaVA-NeXT caption pass, produced by the REFERENCE's vendored model
(llava/model/language_model/llava_llama.py + llava_arch.py + mm_utils.py + conversation.py) on a tiny seeded
LLaMA + CLIP configuration, CPU fp32.  Authoring container only:
    python tests/golden/gen_llava_golden.py        -> tests/golden/llava_next.npz

The reference was written against transformers 4.43; with the installed 5.x three helpers it imports have moved or gone
(apply_chunking_to_forward, prune_linear_layer -> transformers.pytorch_utils; find_pruneable_heads_and_indices removed, only
referenced by resampler code this model does not build) and image processors expose SizeDict objects: both are bridged
here at import level (third-party API relocation), no reference file is touched."""
import copy
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")

import transformers.modeling_utils as MU
import transformers.pytorch_utils as PU


def _gone(*a, **k):
    raise NotImplementedError("removed from transformers 5")


for _n in ("apply_chunking_to_forward", "find_pruneable_heads_and_indices", "prune_linear_layer"):
    if not hasattr(MU, _n):
        setattr(MU, _n, getattr(PU, _n, _gone))

import numpy as np
import torch

import llava_common as C
from llava.constants import IMAGE_TOKEN_INDEX
from llava.conversation import conv_templates
from llava.mm_utils import process_images, tokenizer_image_token
from llava.model.language_model.llava_llama import LlavaConfig, LlavaLlamaForCausalLM

torch.set_num_threads(4)


@torch.no_grad()
def main():
    tower_dir = C.save_tiny_clip("/tmp/llava_golden_clip")
    cfg = LlavaConfig(**C.LLAMA, **C.MM, mm_vision_tower=tower_dir, attn_implementation="sdpa")
    torch.manual_seed(0)
    m = LlavaLlamaForCausalLM(cfg).eval()
    C.name_seeded_state(m, C.WEIGHT_SEED)
    tok = C.build_tokenizer()
    conv = copy.deepcopy(conv_templates["llava_llama_3"])
    conv.tokenizer = tok
    conv.append_message(conv.roles[0], C.QUESTION)
    conv.append_message(conv.roles[1], None)
    prompt = conv.get_prompt()
    ids = tokenizer_image_token(prompt, tok, IMAGE_TOKEN_INDEX, return_tensors="pt").unsqueeze(0)
    proc = C.PlainProcessor(m.get_vision_tower().image_processor)
    out = {"prompt": np.array(prompt), "input_ids": ids.numpy(),
           "param_names": np.array(sorted(k for k, _ in m.named_parameters()))}
    for n, size in enumerate(C.IMAGE_SIZES):
        img = C.test_image(size, 5 + n)
        px = process_images([img], proc, m.config)
        images = [x for x in px]
        r = m.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, images, ["image"], image_sizes=[img.size])
        emb = r[4]
        logits = m(inputs_embeds=emb).logits[0, -1]
        torch.manual_seed(3)
        sampled = m.generate(ids, images=images, image_sizes=[img.size], do_sample=True, temperature=0.2, num_beams=1,
                             max_new_tokens=16, return_dict_in_generate=True, output_scores=True)[0][0]
        greedy = m.generate(ids, images=images, image_sizes=[img.size], do_sample=False, num_beams=1, max_new_tokens=16,
                            return_dict_in_generate=True, output_scores=True)[0][0]
        out[f"i{n}.pixels"], out[f"i{n}.embeds"], out[f"i{n}.logits"] = px[0].numpy(), emb.numpy(), logits.numpy()
        out[f"i{n}.sampled"], out[f"i{n}.greedy"] = sampled.numpy(), greedy.numpy()
        print(f"image {size}: views {tuple(px[0].shape)}, embeds {tuple(emb.shape)}, sampled {sampled.tolist()}, greedy {greedy.tolist()}")
    np.savez_compressed(os.path.join(HERE, "llava_next.npz"), **out)
    print("done")


if __name__ == "__main__":
    main()
