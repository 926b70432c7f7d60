"""BertConfig: plain attribute bag with the reference's field names and defaults
(antmmf/modules/vision/backbone/clip/configuration_bert.py:1-84)."""


class BertConfig:
    def __init__(self, vocab_size_or_config_json_file=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                 intermediate_size=3072, hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
                 max_position_embeddings=512, type_vocab_size=2, initializer_range=0.02, layer_norm_eps=1e-12,
                 output_attentions=False, output_hidden_states=False, **kwargs):
        if isinstance(vocab_size_or_config_json_file, str):
            import json

            with open(vocab_size_or_config_json_file, "r", encoding="utf-8") as f:
                for k, v in json.load(f).items():
                    setattr(self, k, v)
            return
        self.vocab_size = vocab_size_or_config_json_file
        self.hidden_size = hidden_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.intermediate_size = intermediate_size
        self.hidden_act = hidden_act
        self.hidden_dropout_prob = hidden_dropout_prob
        self.attention_probs_dropout_prob = attention_probs_dropout_prob
        self.max_position_embeddings = max_position_embeddings
        self.type_vocab_size = type_vocab_size
        self.initializer_range = initializer_range
        self.layer_norm_eps = layer_norm_eps
        self.output_attentions = output_attentions
        self.output_hidden_states = output_hidden_states
        for k, v in kwargs.items():
            setattr(self, k, v)
