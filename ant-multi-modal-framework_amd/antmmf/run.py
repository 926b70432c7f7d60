"""plain_run (reference: antmmf/run.py:40-63): build config + trainer, trainer.load(); trainer.train()."""
from antmmf.common.build import build_config
from antmmf.trainers.build import build_trainer


def plain_run(args, train_batches=None):
    config = build_config(args.config, getattr(args, "config_override", None), getattr(args, "opts", None))
    trainer = build_trainer(config, train_batches)
    trainer.load()
    trainer.train()
    return trainer
