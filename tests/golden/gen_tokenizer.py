"""Generates tests/golden/tokenizer_synth/tokenizer.json: a small byte-level BPE tokenizer in the BART / Florence-2 layout
(<s>=0, <pad>=1, </s>=2, <unk>=3, ByteLevel pre-tokenizer + decoder), trained on a few GUI-caption sentences.  The real
`microsoft/Florence-2-base` tokenizer files are not on this box (no network); the product reads whatever `tokenizer.json`
sits next to the checkpoint through the same `tokenizers.Tokenizer.from_file` call, which is what this fixture pins."""
from pathlib import Path

from tokenizers import Tokenizer, decoders, models, pre_tokenizers, processors, trainers

CORPUS = [
    "a blue settings icon with a gear", "search button with a magnifying glass", "close window", "a red notification badge",
    "text field for entering a user name", "download arrow", "a folder icon", "play button", "volume slider", "the save icon",
    "a checkbox that is checked", "menu with three horizontal lines", "back arrow", "a user profile picture", "calendar icon",
] * 20

if __name__ == "__main__":
    tok = Tokenizer(models.BPE(unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    trainer = trainers.BpeTrainer(vocab_size=400, special_tokens=["<s>", "<pad>", "</s>", "<unk>"],
                                  initial_alphabet=pre_tokenizers.ByteLevel.alphabet())
    tok.train_from_iterator(CORPUS, trainer)
    tok.post_processor = processors.TemplateProcessing(single="<s> $A </s>", special_tokens=[("<s>", 0), ("</s>", 2)])
    out = Path(__file__).resolve().parent / "tokenizer_synth" / "tokenizer.json"
    tok.save(str(out))
    print(out, tok.get_vocab_size())
